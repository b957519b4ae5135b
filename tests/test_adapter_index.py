"""AdapterIndex path: the dictionary built by cah_index_create against the
reference's (golden fixtures from tests/golden/make_index_golden.py and, when oracle/_ref is
available, the reference's AdapterIndex itself), and the GPU lookup kernel against
IndexedPrefixAdapters / IndexedSuffixAdapters.match_to (reference adapters.py:1289-1567)."""
import io
import random

import numpy as np
import pytest


def _make(A, case):
    cls = A.PrefixAdapter if case["prefix"] else A.SuffixAdapter
    return [cls(s, max_errors=e, indels=bool(i), name=f"a{j}") for j, (s, e, i) in enumerate(case["adapters"])]


def test_index_contents_match_golden(golden):
    """host only: no GPU needed to build and query the dictionary"""
    from cutadapt_amd import adapters as A
    for case in golden("index.json"):
        ix = A.AdapterIndex(_make(A, case), prefix=case["prefix"])
        assert len(ix) == case["n_strings"], case["name"]
        assert ix._lengths == case["lengths"], case["name"]
        assert ix._ambiguous == case["n_ambiguous"], case["name"]
        for line in case["sample"]:
            s, ad, e, m = line.split()
            assert ix._h.get(s) == (int(ad), int(e), int(m)), (case["name"], line)
        assert ix._h.get("ACGTN") is None and ix._h.get("") in (None, ix._h.get(""))


def test_index_contents_match_reference(ref):
    """every entry of the reference's dictionary, for random adapter sets incl. collisions"""
    if ref is None:
        pytest.skip("oracle/_ref not built")
    from cutadapt_amd import adapters as A
    RA = ref.adapters
    rng = random.Random(99)
    compared = 0
    for trial in range(16):
        prefix = trial % 2 == 0
        specs = []
        for _ in range(rng.randint(1, 5)):
            L = rng.randint(3, 12)
            seq = "".join(rng.choice("ACGT") for _ in range(L))
            if specs and rng.random() < 0.3:
                seq = specs[0][0][:L - 1] + rng.choice("ACGT")
            specs.append((seq, rng.choice([0.0, 0.1, 0.2, 0.25, 0.3]), rng.random() < 0.6))
        rcls, mcls = (RA.PrefixAdapter, A.PrefixAdapter) if prefix else (RA.SuffixAdapter, A.SuffixAdapter)
        rads = [rcls(s, max_errors=r, indels=i) for s, r, i in specs]
        ri = RA.AdapterIndex(rads, prefix=prefix)
        mi = A.AdapterIndex([mcls(s, max_errors=r, indels=i) for s, r, i in specs], prefix=prefix)
        assert mi._lengths == ri._lengths and len(mi) == len(ri._index) and mi._ambiguous == ri._ambiguous, specs
        for s, (ad, e, m) in ri._index.items():
            assert mi._h.get(s) == (rads.index(ad), e, m), (specs, s)
        compared += len(ri._index)
    assert compared > 10000


def _wide_specs(rng):
    """adapter sets the 2-bit table cannot hold: more than 60 characters and / or characters other than ACGT
    (adapter_wildcards=False keeps them literal: reference adapters.py:586-588)"""
    specs = []
    kind = rng.randrange(4)
    for _ in range(rng.randint(1, 4)):
        if kind == 0:                                   # long, exact or one error
            L = rng.randint(61, 90)
            seq = "".join(rng.choice("ACGT") for _ in range(L))
            rate = rng.choice([0.0, 0.0, 1.2 / L])
        elif kind == 1:                                 # short with literal non-ACGT characters
            L = rng.randint(4, 14)
            seq = "".join(rng.choice("ACGTNRX") for _ in range(L))
            rate = rng.choice([0.0, 0.1, 0.2, 0.21])
        elif kind == 2:                                 # mixed lengths, one long one
            L = rng.choice([8, 12, 64, 70])
            seq = "".join(rng.choice("ACGT") for _ in range(L))
            rate = rng.choice([0.0, 0.1]) if L < 20 else 0.0
        else:
            L = rng.randint(58, 66)
            seq = "".join(rng.choice("ACGTN") for _ in range(L))
            rate = rng.choice([0.0, 1.2 / L])
        if specs and rng.random() < 0.3:
            seq = specs[0][0][:L - 1] + rng.choice("ACGT")
        specs.append((seq, rate, rng.random() < 0.5))
    return specs


def test_wide_index_contents_match_reference(ref):
    """the hashed-byte-string table: every entry of the reference's dictionary"""
    if ref is None:
        pytest.skip("oracle/_ref not built")
    from cutadapt_amd import adapters as A
    RA = ref.adapters
    rng = random.Random(2024)
    compared = wide = 0
    for trial in range(40):
        prefix = trial % 2 == 0
        specs = _wide_specs(rng)
        rcls, mcls = (RA.PrefixAdapter, A.PrefixAdapter) if prefix else (RA.SuffixAdapter, A.SuffixAdapter)
        rads = [rcls(s, max_errors=r, indels=i, adapter_wildcards=False) for s, r, i in specs]
        mads = [mcls(s, max_errors=r, indels=i, adapter_wildcards=False) for s, r, i in specs]
        assert all(RA.AdapterIndex.is_acceptable(a, prefix) for a in rads)
        assert all(A.AdapterIndex.is_acceptable(a, prefix) for a in mads), specs
        ri = RA.AdapterIndex(rads, prefix=prefix)
        mi = A.AdapterIndex(mads, prefix=prefix)
        assert mi._lengths == ri._lengths and len(mi) == len(ri._index) and mi._ambiguous == ri._ambiguous, specs
        for s, (ad, e, m) in ri._index.items():
            assert mi._h.get(s) == (rads.index(ad), e, m), (specs, s)
        assert mi._h.get("ACGTN") is None or "ACGTN" in ri._index
        compared += len(ri._index)
        wide += any(len(s) > 60 or not set(s) <= set("ACGT") for s, _, _ in specs)
    assert compared > 5000 and wide > 30


def test_index_errors():
    from cutadapt_amd import adapters as A
    from cutadapt_amd import _lib
    with pytest.raises(ValueError, match="Adapter list is empty"):
        A.AdapterIndex([], prefix=True)
    with pytest.raises(ValueError, match="5' anchored"):
        A.AdapterIndex([A.SuffixAdapter("ACGTACGT")], prefix=True)
    with pytest.raises(ValueError, match="3' anchored"):
        A.AdapterIndex([A.PrefixAdapter("ACGTACGT")], prefix=False)
    with pytest.raises(ValueError, match="Error rate too high"):
        A.AdapterIndex([A.PrefixAdapter("ACGTACGTACGTACGTACGT", max_errors=0.2)], prefix=True)
    with pytest.raises(ValueError, match="Wildcards in the read"):
        A.AdapterIndex([A.PrefixAdapter("ACGTACGT", read_wildcards=True)], prefix=True)
    with pytest.raises(ValueError, match="Wildcards in the adapter"):
        A.AdapterIndex([A.PrefixAdapter("ACGTNCGT")], prefix=True)
    assert A.AdapterIndex.is_acceptable(A.PrefixAdapter("ACGTACGT"), prefix=True)
    assert not A.AdapterIndex.is_acceptable(A.PrefixAdapter("ACGTACGT"), prefix=False)
    assert not A.AdapterIndex.is_acceptable(A.BackAdapter("ACGTACGT"), prefix=False)
    # the C ABI reports the same conditions without the Python layer
    with pytest.raises(ValueError, match="Error rate too high"):
        _lib.Index([("ACGTACGTACGTACGTACGT", 0.2, True)], True)
    with pytest.raises(_lib.UnsupportedByHipPath):
        _lib.Index([("ACGT" * 251, 0.0, True)], True)
    assert not A.AdapterIndex.is_acceptable(A.PrefixAdapter("ACGT" * 251, max_errors=0), prefix=True)
    with pytest.raises(ValueError, match="Adapter list is empty"):
        _lib.Index([], True)


def test_cutter_regroups_indexable_adapters():
    """AdapterCutter._regroup_into_indexed_adapters (reference modifiers.py:124-141)"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.pipeline import BatchAdapterCutter
    ads = [A.PrefixAdapter("ACGTACGT", name="p1"), A.BackAdapter("TTTTGGGG", name="b"),
           A.PrefixAdapter("GGGGACGT", name="p2"), A.SuffixAdapter("CCCCAAAA", name="s1")]
    c = BatchAdapterCutter(ads)
    assert [a.name for a in c.all_adapters] == ["b", "p1", "p2", "s1"]
    kinds = [(k, type(u).__name__) for k, u, _ in c._units]
    assert kinds == [("fused", "MultipleAdapters"), ("fused", "IndexedPrefixAdapters"), ("fused", "MultipleAdapters")]
    c = BatchAdapterCutter(ads, index=False)
    assert [a.name for a in c.all_adapters] == ["p1", "b", "p2", "s1"] and len(c._units) == 1
    c = BatchAdapterCutter(ads[:2])                       # a single anchored adapter: order untouched
    assert [a.name for a in c.all_adapters] == ["p1", "b"]


@pytest.mark.gpu
def test_index_lookup_golden(hip, golden):
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    for case in golden("index.json"):
        ads = _make(A, case)
        matcher = (A.IndexedPrefixAdapters if case["prefix"] else A.IndexedSuffixAdapters)(ads)
        ascii_reads = [r for r in case["reads"]]
        bm = matcher.match_to_batch(ReadBatch.from_strings(ascii_reads))
        for i, (read, exp) in enumerate(zip(case["reads"], case["results"])):
            got = None if not bm.found[i] else [int(bm.adapter_index[i])] + [int(v) for v in bm.coords[i]]
            assert got == exp, (case["name"], read, got, exp)
            m = matcher.match_to(read)                    # batch of one through the host entry point
            one = None if m is None else [ads.index(m.adapter), m.astart, m.astop, m.rstart, m.rstop, m.score, m.errors]
            assert one == exp, (case["name"], read, one, exp)
            if m is not None:
                assert type(m).__name__ == ("RemoveBeforeMatch" if case["prefix"] else "RemoveAfterMatch")


@pytest.mark.gpu
def test_index_lookup_fuzz_against_reference(hip, ref):
    """random adapter sets x random reads (hits with 0..3 edits, N, lower case, short reads)
    against the reference's IndexedPrefix/SuffixAdapters.match_to"""
    assert ref is not None, "oracle/_ref must travel to the GPU box"
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    RA = ref.adapters
    rng = random.Random(4242)
    total = matched = 0
    for trial in range(24):
        prefix = trial % 2 == 0
        n_ad = rng.randint(2, 20)
        base_len = rng.randint(5, 16)
        specs = []
        for _ in range(n_ad):
            L = base_len + rng.choice([0, 0, 0, 1, -1, 2])
            specs.append(("".join(rng.choice("ACGT") for _ in range(L)), rng.choice([0.1, 0.15, 0.2, 0.25]),
                          rng.random() < 0.6))
        rcls, mcls = (RA.PrefixAdapter, A.PrefixAdapter) if prefix else (RA.SuffixAdapter, A.SuffixAdapter)
        try:
            rm = (RA.IndexedPrefixAdapters if prefix else RA.IndexedSuffixAdapters)(
                [rcls(s, max_errors=e, indels=i) for s, e, i in specs])
        except ValueError:
            continue
        rads = rm._index._adapters
        mm = (A.IndexedPrefixAdapters if prefix else A.IndexedSuffixAdapters)(
            [mcls(s, max_errors=e, indels=i) for s, e, i in specs])
        reads = []
        for _ in range(1500):
            s = list(rng.choice(specs)[0])
            for _ in range(rng.choice([0, 0, 1, 1, 2, 3])):
                op, p = rng.choice("sid"), rng.randrange(len(s) + 1)
                if op == "s" and s:
                    s[min(p, len(s) - 1)] = rng.choice("ACGTN")
                elif op == "i":
                    s.insert(p, rng.choice("ACGTN"))
                elif s:
                    del s[min(p, len(s) - 1)]
            pad = "".join(rng.choice("ACGTN" if rng.random() < 0.1 else "ACGT") for _ in range(rng.randint(0, 20)))
            r = "".join(s) + pad if prefix else pad + "".join(s)
            if rng.random() < 0.1:
                r = r.lower()
            if rng.random() < 0.1:
                cut = rng.randint(0, len(r))
                r = r[:cut] if prefix else r[cut:]
            reads.append(r)
        bm = mm.match_to_batch(ReadBatch.from_strings(reads))
        for i, r in enumerate(reads):
            m = rm.match_to(r)
            exp = None if m is None else (rads.index(m.adapter), m.astart, m.astop, m.rstart, m.rstop, m.score, m.errors)
            got = None if not bm.found[i] else (int(bm.adapter_index[i]),) + tuple(int(v) for v in bm.coords[i])
            assert got == exp, (specs, r, got, exp)
            matched += exp is not None
        total += len(reads)
    assert total > 20000 and matched > 5000


@pytest.mark.gpu
def test_demultiplex_pipeline_with_index_matches_unindexed(hip):
    """AdapterCutter(index=True) against the same adapters matched one by one (index=False): the
    reference guarantees identical trimming whenever no read is ambiguous between barcodes"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.pipeline import BatchAdapterCutter, read_fastq_chunks
    rng = random.Random(8)
    barcodes = []
    while len(barcodes) < 24:                             # pairwise Hamming distance >= 4: no ambiguity at k = 1
        b = "".join(rng.choice("ACGT") for _ in range(10))
        if all(sum(x != y for x, y in zip(b, o)) >= 4 for o in barcodes):
            barcodes.append(b)
    reads = []
    for i in range(3000):
        b = list(rng.choice(barcodes))
        if rng.random() < 0.3:
            b[rng.randrange(10)] = rng.choice("ACGT")
        body = "".join(rng.choice("ACGT") for _ in range(rng.randint(20, 60)))
        reads.append(("".join(b) if rng.random() < 0.9 else "") + body)
    fq = "".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(reads)).encode()
    chunk = list(read_fastq_chunks(io.BytesIO(fq)))[0]
    seqs, offsets = chunk.pack_sequences()
    res = {}
    for index in (True, False):
        ads = [A.PrefixAdapter(b, max_errors=0.1, indels=False, name=f"bc{j}") for j, b in enumerate(barcodes)]
        cutter = BatchAdapterCutter(ads, index=index)
        res[index] = cutter.process_arrays(seqs, offsets)
        assert cutter.with_adapters > 2000
    for key in ("beg", "end", "matched"):
        assert np.array_equal(res[True][key], res[False][key]), key
    assert np.array_equal(res[True]["rows"], res[False]["rows"])


@pytest.mark.gpu
def test_wide_index_lookup_fuzz_against_reference(hip, ref):
    """adapters of more than 60 characters and / or with literal non-ACGT characters (the hashed-byte-string
    table, k_index_lookup_wide) against the reference's IndexedPrefix/SuffixAdapters.match_to: hits with 0..3
    edits, N (re-alignment with the columns in LDS), the adapters' own odd characters, lower case, short reads"""
    assert ref is not None, "oracle/_ref must travel to the GPU box"
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    RA = ref.adapters
    rng = random.Random(777)
    total = matched = with_n = 0
    for trial in range(40):
        prefix = trial % 2 == 0
        specs = _wide_specs(rng)
        rcls, mcls = (RA.PrefixAdapter, A.PrefixAdapter) if prefix else (RA.SuffixAdapter, A.SuffixAdapter)
        rm = (RA.IndexedPrefixAdapters if prefix else RA.IndexedSuffixAdapters)(
            [rcls(s, max_errors=e, indels=i, adapter_wildcards=False) for s, e, i in specs])
        rads = rm._index._adapters
        mm = (A.IndexedPrefixAdapters if prefix else A.IndexedSuffixAdapters)(
            [mcls(s, max_errors=e, indels=i, adapter_wildcards=False) for s, e, i in specs])
        reads = []
        for _ in range(800):
            s = list(rng.choice(specs)[0])
            for _ in range(rng.choice([0, 0, 0, 1, 1, 2, 3])):
                op, p = rng.choice("ssid"), rng.randrange(len(s) + 1)
                if op == "s" and s:
                    s[min(p, len(s) - 1)] = rng.choice("ACGTNRX")
                elif op == "i":
                    s.insert(p, rng.choice("ACGTN"))
                elif s:
                    del s[min(p, len(s) - 1)]
            pad = "".join(rng.choice("ACGTN" if rng.random() < 0.1 else "ACGT") for _ in range(rng.randint(0, 20)))
            r = "".join(s) + pad if prefix else pad + "".join(s)
            if rng.random() < 0.1:
                r = r.lower()
            if rng.random() < 0.1:
                cut = rng.randint(0, len(r))
                r = r[:cut] if prefix else r[cut:]
            reads.append(r)
        reads += ["", "A", "N"]
        bm = mm.match_to_batch(ReadBatch.from_strings(reads))
        for i, r in enumerate(reads):
            m = rm.match_to(r)
            exp = None if m is None else (rads.index(m.adapter), m.astart, m.astop, m.rstart, m.rstop, m.score, m.errors)
            got = None if not bm.found[i] else (int(bm.adapter_index[i]),) + tuple(int(v) for v in bm.coords[i])
            assert got == exp, (specs, r, got, exp)
            matched += exp is not None
            with_n += exp is not None and "N" in r.upper()
        for r in reads[:20]:                                # the per-read entry point
            m, e = mm.match_to(r), rm.match_to(r)
            assert (m is None) == (e is None) and (m is None or (m.rstart, m.rstop, m.score, m.errors) ==
                                                   (e.rstart, e.rstop, e.score, e.errors)), (specs, r)
        total += len(reads)
    assert total > 20000 and matched > 5000 and with_n > 200, (total, matched, with_n)
