"""Constructors of Aligner / PrefixComparer / SuffixComparer (cutadapt_amd.align) against the reference's compiled classes for
500 random parameter sets, valid and not (tests/golden/make_ctor_golden.py, reference _align.pyx:200-277, :607-640): the
exception class, repr(), effective_length, the pickle arguments.  No GPU: plans are built on the host."""
import json
import os

import pytest

from cutadapt_amd import align as M

HERE = os.path.dirname(os.path.abspath(__file__))


def test_constructors_accept_refuse_and_describe_as_the_reference():
    with open(os.path.join(HERE, "golden", "ctors.json")) as f:
        cases = json.load(f)
    built = refused = 0
    for c in cases:
        cls = getattr(M, c["cls"])
        ctx = (c["cls"], c["args"])
        if "error" in c:
            with pytest.raises(Exception) as info:
                cls(*c["args"])
            assert type(info.value).__name__ == c["error"], (ctx, repr(info.value))
            refused += 1
            continue
        obj = cls(*c["args"])
        assert repr(obj) == c["repr"], ctx
        if "reduce" in c:
            red = obj.__reduce__()
            assert [red[0].__name__, list(red[1])] == c["reduce"], ctx
            again = red[0](*red[1])
            assert repr(again) == c["repr"], ctx
        if "effective_length" in c:
            assert obj.effective_length == c["effective_length"], ctx
        built += 1
    assert built > 150 and refused > 150
