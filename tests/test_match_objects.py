"""The Match objects of the adapter API (cutadapt_amd.adapters: RemoveBeforeMatch, RemoveAfterMatch, LinkedMatch, remainder)
against golden vectors taken from the reference's own classes (tests/golden/make_match_golden.py, reference adapters.py:292-493,
:1092-1153, :1588-1602).  No GPU: the objects are plain data over an alignment tuple -- what the modifiers and the report
read off a match must be the reference's, whatever produced the tuple."""
import json
import os
import types

import pytest

from cutadapt_amd import adapters as A

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "matches.json")) as f:
        return json.load(f)


def test_single_matches_answer_as_the_reference(golden):
    adapters = {}
    seen = set()
    for c in golden["single"]:
        key = (c["before"], c["adapter"])
        if key not in adapters:
            adapters[key] = (A.FrontAdapter if c["before"] else A.BackAdapter)(c["adapter"], name="ad")
        cls = A.RemoveBeforeMatch if c["before"] else A.RemoveAfterMatch
        m = cls(*c["tuple"], adapters[key], c["sequence"])
        read = types.SimpleNamespace(sequence=c["sequence"], qualities=c["qualities"])
        sl = m.trim_slice()
        assert repr(m) == c["repr"]
        assert m.length == c["length"] and list(m.astuple()) == c["tuple"]
        assert list(m.remainder_interval()) == c["remainder_interval"]
        assert list(m.retained_adapter_interval()) == c["retained_adapter_interval"]
        assert [sl.start, sl.stop, sl.step] == c["trim_slice"]
        assert m.trimmed(c["sequence"]) == c["trimmed"] and m.rest() == c["rest"]
        assert m.removed_sequence_length() == c["removed_sequence_length"]
        assert m.match_sequence() == c["match_sequence"] and m.wildcards() == c["wildcards"]
        assert m.get_info_records(read) == c["info"]
        if not c["before"]:
            assert m.adjacent_base() == c["adjacent_base"]
        assert m == cls(*c["tuple"], adapters[key], c["sequence"])
        other = (A.RemoveAfterMatch if c["before"] else A.RemoveBeforeMatch)(*c["tuple"], adapters[key], c["sequence"])
        assert not (m == other)
        seen.add((c["before"], c["qualities"] is None, bool(c["wildcards"])))
    assert len(seen) >= 7                                     # both sides, with and without qualities, with and without wildcards


def test_linked_matches_answer_as_the_reference(golden):
    fa, ba = A.FrontAdapter("ACGTA", name="f"), A.BackAdapter("TGCAT", name="b")
    la = A.LinkedAdapter(fa, ba, front_required=True, back_required=False, name="la")
    shapes = set()
    for c in golden["linked"]:
        seq = c["sequence"]
        front = A.RemoveBeforeMatch(*c["front"], fa, seq) if "front" in c else None
        rest = seq[c["front"][3]:] if front is not None else seq
        back = A.RemoveAfterMatch(*c["back"], ba, rest) if "back" in c else None
        lm = A.LinkedMatch(front, back, la)
        assert (lm.score, lm.errors) == (c["score"], c["errors"])
        assert lm.trimmed(seq) == c["trimmed"]
        assert list(lm.remainder_interval()) == c["remainder_interval"]
        assert list(lm.retained_adapter_interval()) == c["retained_adapter_interval"]
        assert lm.match_sequence() == c["match_sequence"]
        assert list(A.remainder([m for m in (front, back) if m is not None])) == c["remainder_of_parts"]
        shapes.add((front is not None, back is not None))
    assert shapes == {(True, True), (True, False), (False, True)}
    with pytest.raises(ValueError):
        A.remainder([])


def test_match_base_classes_are_abstract():
    with pytest.raises(TypeError):
        A.Match()
    with pytest.raises(TypeError):
        A.SingleMatch(0, 1, 0, 1, 1, 0, A.BackAdapter("ACGT"), "A")
