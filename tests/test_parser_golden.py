"""The adapter-specification grammar (cutadapt_amd.pipeline.adapter_from_spec: -a / -g / -b SPEC) against what the reference's
own parser builds or refuses for 900 random specifications (tests/golden/make_parser_golden.py, reference parser.py:28-151,
:441-551): names, anchors, X markers, brace repeats, ellipsis forms, linked adapters, every search parameter.  No GPU."""
import json
import os

import pytest

from cutadapt_amd import adapters as A
from cutadapt_amd.pipeline import adapter_from_spec

HERE = os.path.dirname(os.path.abspath(__file__))


def describe_single(ad):
    return {"cls": type(ad).__name__, "sequence": ad.sequence, "max_error_rate": float(ad.max_error_rate),
            "min_overlap": ad.min_overlap, "indels": bool(ad.indels), "read_wildcards": bool(ad.read_wildcards),
            "adapter_wildcards": bool(ad.adapter_wildcards), "spec": ad.spec(),
            "flags": int(ad.aligner.flags) if hasattr(ad.aligner, "flags") else None, "aligner": type(ad.aligner).__name__}


def test_specifications_build_what_the_reference_builds():
    with open(os.path.join(HERE, "golden", "parser.json")) as f:
        g = json.load(f)
    defaults = g["defaults"]
    built = refused = 0
    for c in g["cases"]:
        ctx = (c["spec"], c["type"])
        if "error" in c:
            with pytest.raises(Exception) as info:
                adapter_from_spec(c["spec"], c["type"], **defaults)
            assert type(info.value).__name__ == c["error"], (ctx, repr(info.value))
            refused += 1
            continue
        ad = adapter_from_spec(c["spec"], c["type"], **defaults)
        w = dict(c["want"])
        name = w.pop("name", None)
        if name is not None:
            assert ad.name == name, ctx
        if w["cls"] == "LinkedAdapter":
            assert type(ad) is A.LinkedAdapter, ctx
            assert (bool(ad.front_required), bool(ad.back_required)) == (w["front_required"], w["back_required"]), ctx
            assert describe_single(ad.front_adapter) == w["front"], ctx
            assert describe_single(ad.back_adapter) == w["back"], ctx
        else:
            assert describe_single(ad) == w, ctx
        built += 1
    assert built > 500 and refused > 250
