"""Adapter.match_to with the ORACLE in the kernels' place: the host logic around the kernels -- which aligner flags and
k-mer search sets a class builds, reversal and mirroring of the rightmost types, which Match type a hit becomes, the two
stages of a linked adapter, the best-of rule of MultipleAdapters -- replayed on the CPU against results taken from the
reference's own classes (tests/golden/adapters.json, linked_multiple.json: the files the -m gpu tests use, and
adapters_extreme.json: the edges of the parameter space, tests/golden/make_adapters_extreme_golden.py).

The one library call of SingleAdapter.match_to (_locate_fused: kmers_present -> locate of the adapter's fused plan) is
replaced by the same two steps of the oracle on the adapter's matcher_spec(); nothing else is patched.  The kernels
themselves are compared with the oracle by the -m gpu tests."""
import json
import os

import pytest

from cutadapt_amd import adapters as A
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def golden(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


@pytest.fixture
def oracle_kernels(monkeypatch):
    cache = {}

    def locate_fused(self, sequence):
        if id(self) not in cache:
            spec = self.matcher_spec()
            if spec.kind == 0:
                engine = orc.Aligner(spec.sequence, spec.max_error_rate, spec.flags, spec.wildcard_ref, spec.wildcard_query,
                                     spec.indel_cost, spec.min_overlap)
            else:
                cls = orc.PrefixComparer if spec.kind == 1 else orc.SuffixComparer
                engine = cls(spec.sequence, spec.max_error_rate, spec.wildcard_ref, spec.wildcard_query, spec.min_overlap)
            finder = None
            if spec.kmer_sets is not None:
                # (the search sets as the adapter's finder holds them -- empty k-mers included, as the reference's would)
                finder = orc.KmerFinder(self.kmer_finder.positions_and_kmers, spec.kmer_ref_wildcards, spec.kmer_query_wildcards)
            cache[id(self)] = (self, engine, finder)                     # (keeps `self` alive: ids are not reused)
        _, engine, finder = cache[id(self)]
        if finder is not None and not finder.kmers_present(sequence):
            return None
        return engine.locate(sequence)

    monkeypatch.setattr(A.SingleAdapter, "_locate_fused", locate_fused)
    return cache


def shown(match):
    return None if match is None else {"cls": type(match).__name__, "t": list(match.astuple())}


@pytest.mark.parametrize("name", ["adapters.json", "adapters_extreme.json"])
def test_every_adapter_class_matches_as_the_reference(oracle_kernels, name):
    n = hits = 0
    kinds = set()
    for c in golden(name):
        adapter = getattr(A, c["cls"])(c["sequence"], **c["kwargs"])
        for read, want in c["reads"]:
            got = shown(adapter.match_to(read))
            assert got == want, (c["cls"], c["sequence"], c["kwargs"], read, got, want)
            n += 1
            hits += want is not None
        kinds.add((c["cls"], bool(c["kwargs"].get("force_anywhere"))))
    assert n >= 2000 and hits >= 200 and len({k[0] for k in kinds}) == 9
    if name == "adapters_extreme.json":
        assert sum(1 for k in kinds if k[1]) == 8                         # force_anywhere on every class that takes it


def test_linked_and_multiple_adapters_match_as_the_reference(oracle_kernels):
    g = golden("linked_multiple.json")
    n = 0
    for c in g["linked"]:
        front = getattr(A, c["front_cls"])(c["front"], max_errors=0.1)         # (the parameters the golden was made with)
        back = A.BackAdapter(c["back"], max_errors=0.1, min_overlap=3)
        linked = A.LinkedAdapter(front, back, c["front_required"], c["back_required"], name="linked")
        for read, want in c["reads"]:
            m = linked.match_to(read)
            got = None if m is None else {"front": shown(m.front_match), "back": shown(m.back_match)}
            assert got == want, (c["front"], c["back"], read, got, want)
            n += 1
    for c in g["multiple"]:
        ads = [A.BackAdapter(s, max_errors=0.15, min_overlap=3) for s in c["seqs"]]
        multi = A.MultipleAdapters(ads)
        for read, want in c["reads"]:
            m = multi.match_to(read)
            got = None if m is None else {"adapter": ads.index(m.adapter), "m": shown(m)}
            assert got == want, (c["seqs"], read, got, want)
            n += 1
    assert n > 500
