"""The nine adapter classes before any read is matched (cutadapt_amd.adapters: one table row per class) against what the
reference's own classes say about themselves (tests/golden/make_adapter_attr_golden.py, reference adapters.py:496-1089):
normalised parameters, names and specifications, which aligner and which k-mer search sets they build -- the
`force_anywhere` parts of linked adapters and the anchored classes without indels included.  No GPU: construction is host
work (the plan builder of the library)."""
import json
import os
import re

import pytest

from cutadapt_amd import adapters as A

HERE = os.path.dirname(os.path.abspath(__file__))


def _norm(text):
    """oracle/_ref holds adapters.py COMPILED (Cython's pure-Python mode): there the annotation `max_error_rate: float` makes
    an integer 0 print as 0.0; the interpreted reference -- and this package -- print what was passed"""
    return re.sub(r"max_error_rate=([0-9.e-]+)", lambda m: "max_error_rate=%r" % float(m.group(1)), text)


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(HERE, "golden", "adapter_attrs.json")) as f:
        return json.load(f)


def test_adapter_classes_describe_themselves_as_the_reference(cases):
    seen = set()
    indexable = 0
    for c in cases:
        cls = getattr(A, c["cls"])
        if "error" in c:
            with pytest.raises(Exception) as info:
                cls(c["seq"], **c["kwargs"])
            assert type(info.value).__name__ == c["error"], c
            continue
        ad = cls(c["seq"], **c["kwargs"])
        w = c["want"]
        ctx = (c["cls"], c["seq"], c["kwargs"])
        assert ad.sequence == w["sequence"] and ad.min_overlap == w["min_overlap"], ctx
        assert ad.max_error_rate == pytest.approx(w["max_error_rate"], abs=0, rel=1e-15), ctx
        flags = (bool(ad.read_wildcards), bool(ad.adapter_wildcards), bool(ad.indels))
        assert flags == (w["read_wildcards"], w["adapter_wildcards"], w["indels"]), ctx
        assert ad.description == w["description"] and ad.spec() == w["spec"], ctx
        assert ad.descriptive_identifier() == w["identifier"] and _norm(repr(ad)) == _norm(w["repr"]), ctx
        assert len(ad) == w["len"] and ad.effective_length == w["effective_length"], ctx
        assert bool(ad.allows_partial_matches) == w["allows_partial_matches"], ctx
        assert type(ad.aligner).__name__ == w["aligner"] and type(ad.kmer_finder).__name__ == w["finder"], ctx
        if "aligner_args" in w:
            assert list(ad.aligner.__reduce__()[1]) == w["aligner_args"], ctx
        if w["finder"] == "KmerFinder":
            # (the reference collects k-mers in Python sets: the order of the search sets and of their k-mers follows
            # the process's string hashing -- compared as collections)
            def key(x):
                return (x[0], -1 if x[1] is None else x[1], x[2])
            got = [[a, b, sorted(k)] for a, b, k in ad.kmer_finder.positions_and_kmers]
            assert sorted(got, key=key) == sorted(w["kmer_sets"], key=key), ctx
            assert [bool(ad.kmer_finder.ref_wildcards), bool(ad.kmer_finder.query_wildcards)] == w["kmer_wildcards"], ctx
        assert [A.AdapterIndex.is_acceptable(ad, True), A.AdapterIndex.is_acceptable(ad, False)] == w["indexable"], ctx
        indexable += any(w["indexable"])
        seen.add((c["cls"], w["aligner"], bool(c["kwargs"].get("force_anywhere"))))
    assert {s[0] for s in seen} == {"FrontAdapter", "RightmostFrontAdapter", "BackAdapter", "RightmostBackAdapter",
                                    "AnywhereAdapter", "NonInternalFrontAdapter", "NonInternalBackAdapter",
                                    "PrefixAdapter", "SuffixAdapter"}
    assert any(s[1] == "PrefixComparer" for s in seen) and any(s[1] == "SuffixComparer" for s in seen)
    assert indexable >= 8                                    # ... and the index of anchored adapters takes some of them
    assert sum(1 for s in seen if s[2]) >= 8                 # force_anywhere seen for every class that takes it


def test_class_hierarchy_is_the_reference_s():
    """isinstance checks elsewhere go by it (AdapterIndex, LinkedAdapter, the specification parser, the modifiers)"""
    assert issubclass(A.RightmostFrontAdapter, A.FrontAdapter) and issubclass(A.RightmostBackAdapter, A.BackAdapter)
    assert issubclass(A.NonInternalFrontAdapter, A.FrontAdapter) and issubclass(A.NonInternalBackAdapter, A.BackAdapter)
    assert issubclass(A.PrefixAdapter, A.NonInternalFrontAdapter) and issubclass(A.SuffixAdapter, A.NonInternalBackAdapter)
    assert not issubclass(A.AnywhereAdapter, (A.FrontAdapter, A.BackAdapter)) and issubclass(A.AnywhereAdapter, A.SingleAdapter)
