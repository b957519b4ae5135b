"""Parity of the HIP path with the oracle and the reference's golden vectors (GPU only).

Every test goes through the C ABI of libcutadapt_hip.so (ctypes -> cah_*), either directly
or via the Python mirror classes.  Bit-exact: all results are integers.
"""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BACK, FRONT, PREFIX, SUFFIX, ANYWHERE = 14, 11, 8, 2, 15
TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
ALPHABETS = ["ACGT", "ACGTN", "ACGTNRYacgtn", "ACGTXNSWKMBDHVU"]


def rs(rng, n, al):
    return "".join(rng.choice(al) for _ in range(n))


def host_locate_batch(plan, seqs, offsets, adapter=0):
    from cutadapt_amd import _lib
    n = len(offsets) - 1
    out6 = np.zeros((n, 6), dtype=np.int32)
    status = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.lib().cah_locate_batch_host(plan.handle, adapter, seqs.ctypes.data, offsets.ctypes.data,
                                                n, out6.ctypes.data, status.ctypes.data))
    return out6, status


def host_match_batch(plan, seqs, offsets):
    from cutadapt_amd import _lib
    n = len(offsets) - 1
    out6 = np.zeros((n, 6), dtype=np.int32)
    status = np.zeros(n, dtype=np.uint8)
    best = np.zeros(n, dtype=np.int32)
    _lib.check(_lib.lib().cah_match_batch_host(plan.handle, seqs.ctypes.data, offsets.ctypes.data, n,
                                               out6.ctypes.data, best.ctypes.data, status.ctypes.data))
    return out6, status, best


def assert_same(out_a, st_a, out_b, st_b, what=""):
    assert np.array_equal(st_a, st_b), f"{what}: status differs at {np.nonzero(st_a != st_b)[0][:10]}"
    bad = np.nonzero((out_a != out_b).any(axis=1))[0]
    assert len(bad) == 0, f"{what}: tuples differ at {bad[:10]}: {out_a[bad[:3]]} vs {out_b[bad[:3]]}"


# ---------------------------------------------------------------------------------------------
# Aligner.locate
# ---------------------------------------------------------------------------------------------
def test_known_answers_through_python_api(hip):
    from cutadapt_amd.align import Aligner
    assert Aligner("", 0, flags=0, min_overlap=0).locate("") == (0, 0, 0, 0, 0, 0)
    assert Aligner("CCAGTCCTCT", 0.3, flags=PREFIX).locate("CCAGTCCTTTCCTGAGAGT") == (0, 10, 0, 10, 8, 1)
    assert Aligner("TCGATC", 1.5 / 6, flags=PREFIX).locate("TCGATGC") == (0, 6, 0, 6, 4, 1)
    assert Aligner("GCCGAACTTCTTAGACTGCCTTAAGGACGT", 0.1, flags=BACK).locate(
        "CAAATCACCAGAAGGCGCCTAACTTCTTAGACTGCC") == (0, 20, 16, 36, 18, 1)
    assert Aligner("TTTT", 0.25, flags=BACK).locate("CCTTTT") == (0, 4, 2, 6, 4, 0)
    assert Aligner("TTTTTT", 0.25, flags=BACK).locate("CCTTTT") == (0, 4, 2, 6, 4, 0)
    assert Aligner("A" * 17, 0.0, BACK).locate("ACAG" + "A" * 42) == (0, 17, 4, 21, 17, 0)
    assert Aligner("CTGATCTGGCCG", 0.1, BACK).locate("AAAAGGG") is None
    a = Aligner("AGGNNNNNNNNNNNNNNTTC", 0.1, BACK, wildcard_ref=True, min_overlap=3)
    assert a.effective_length == 6
    assert a.locate("TTC") is None
    assert a.locate("AGGCCCCCCC")[:4] == (0, 10, 0, 10)
    with pytest.raises(ValueError):
        Aligner("ACGT", 0.1).locate("ACéT")
    assert "Aligner(reference='ACGT'" in repr(Aligner("ACGT", 0.1))
    import pickle
    b = pickle.loads(pickle.dumps(Aligner("TTTT", 0.25, flags=BACK)))
    assert b.locate("CCTTTT") == (0, 4, 2, 6, 4, 0)


def test_golden_locate(hip, golden):
    """reference outputs for all flag combinations / wildcard modes / indel costs"""
    from cutadapt_amd import _lib
    cases = golden("locate.json")
    groups = {}
    for c in cases:
        key = (c["ref"], c["rate"], c["flags"], c["wr"], c["wq"], c["indel_cost"], c["min_overlap"])
        groups.setdefault(key, []).append(c)
    from oracle.oracle import pack_reads
    n_checked = 0
    for key, cs in groups.items():
        plan = _lib.Plan([_lib.MatcherSpec(*key)])
        assert plan.effective_length(0) == cs[0]["effective_length"]
        seqs, offsets = pack_reads([c["query"] for c in cs])
        out6, status = host_locate_batch(plan, seqs, offsets)
        for i, c in enumerate(cs):
            if c["result"] is None:
                assert status[i] == 0, (key, c["query"], out6[i])
            else:
                assert status[i] == 1 and out6[i].tolist() == c["result"], (key, c["query"], out6[i], c["result"])
            n_checked += 1
    assert n_checked == len(cases)


def test_golden_truseq(hip, golden):
    from cutadapt_amd import _lib
    from oracle.oracle import pack_reads
    g = golden("truseq.json")
    plan = _lib.Plan([_lib.MatcherSpec(g["ref"], g["rate"], g["flags"], False, False, 1, g["min_overlap"])])
    seqs, offsets = pack_reads([c["query"] for c in g["cases"]])
    out6, status = host_locate_batch(plan, seqs, offsets)
    for i, c in enumerate(g["cases"]):
        if c["result"] is None:
            assert status[i] == 0
        else:
            assert status[i] == 1 and out6[i].tolist() == c["result"], (c["query"], out6[i], c["result"])


def test_fuzz_locate_vs_oracle(hip, orc):
    """randomized aligners (all 16 flag combos, wildcards, indel costs, m up to 64) on batches
    of ragged reads incl. empty ones"""
    from cutadapt_amd import _lib
    rng = random.Random(4242)
    total = 0
    for it in range(160):
        al = rng.choice(ALPHABETS)
        m = rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 25, 31, 32, 33, 40, 41, 47, 48, 55, 56, 57, 63, 64])
        adapter = rs(rng, m, al)
        args = (adapter, rng.choice([0, 0.05, 0.1, 0.2, 0.3, 0.5, 1.0, rng.random()]), rng.randint(0, 15),
                rng.random() < 0.3, rng.random() < 0.3, rng.choice([1, 1, 2, 100000]), rng.randint(1, min(m, 6)))
        try:
            oa = orc.Aligner(*args)
        except ValueError:
            with pytest.raises(ValueError):
                _lib.Plan([_lib.MatcherSpec(*args)])
            continue
        plan = _lib.Plan([_lib.MatcherSpec(*args)])
        reads = []
        for _ in range(300):
            n = rng.choice([0, 1, 2, rng.randint(0, 60), rng.randint(0, 200)])
            q = rs(rng, n, al)
            if rng.random() < 0.6 and n:
                p = rng.randint(0, n)
                piece = list(adapter[rng.randint(0, m - 1):][:rng.randint(1, m)])
                for _ in range(rng.randint(0, 3)):
                    if piece:
                        x = rng.randrange(len(piece))
                        op = rng.randint(0, 2)
                        if op == 0:
                            piece[x] = rng.choice(al)
                        elif op == 1:
                            piece.insert(x, rng.choice(al))
                        else:
                            del piece[x]
                q = q[:p] + "".join(piece) + q[p:]
            reads.append(q)
        seqs, offsets = orc.pack_reads(reads)
        want6, want_st = oa.locate_batch(seqs, offsets)
        got6, got_st = host_locate_batch(plan, seqs, offsets)
        assert_same(got6, got_st, want6, want_st, f"aligner {args}")
        total += len(reads)
    assert total > 30000


def test_invalid_bytes_are_flagged(hip):
    from cutadapt_amd import _lib
    plan = _lib.Plan([_lib.MatcherSpec("ACGTACGT", 0.1, BACK)])
    seqs = np.frombuffer(b"TTACGTACGT" + b"TTAC\xc3\xa9GTAC" + b"ACGTACGTAA", dtype=np.uint8).copy()
    offsets = np.array([0, 10, 20, 30], dtype=np.int64)
    out6, status = host_locate_batch(plan, seqs, offsets)
    assert status.tolist() == [1, 2, 1]


# ---------------------------------------------------------------------------------------------
# comparers
# ---------------------------------------------------------------------------------------------
def test_golden_comparers(hip, golden):
    from cutadapt_amd.align import PrefixComparer, SuffixComparer
    cache = {}
    for c in golden("comparers.json"):
        key = (c["kind"], c["ref"], c["rate"], c["wr"], c["wq"], c["min_overlap"])
        if key not in cache:
            cls = PrefixComparer if c["kind"] == "prefix" else SuffixComparer
            cache[key] = cls(c["ref"], c["rate"], c["wr"], c["wq"], c["min_overlap"])
        cmp_ = cache[key]
        assert cmp_.effective_length == c["effective_length"]
        want = tuple(c["result"]) if c["result"] is not None else None
        assert cmp_.locate(c["query"]) == want, c


def test_comparer_errors(hip):
    from cutadapt_amd.align import PrefixComparer, SuffixComparer
    with pytest.raises(ValueError):
        PrefixComparer("NNN", 0.1, wildcard_ref=True)
    with pytest.raises(ValueError):
        PrefixComparer("ACGT", 1.5)
    with pytest.raises(ValueError):
        SuffixComparer("ACGT", 0.1, min_overlap=0)
    assert "PrefixComparer(" in repr(PrefixComparer("ACGT", 0.5))


# ---------------------------------------------------------------------------------------------
# KmerFinder
# ---------------------------------------------------------------------------------------------
def test_golden_kmers(hip, golden):
    from cutadapt_amd._kmer_finder import KmerFinder
    from cutadapt_amd import _lib
    from oracle.oracle import pack_reads
    for c in golden("kmers.json"):
        f = KmerFinder([(a, b, k) for a, b, k in c["sets"]], c["wr"], c["wq"])
        seqs, offsets = pack_reads([r for r, _ in c["reads"]])
        present = np.zeros(len(c["reads"]), dtype=np.uint8)
        _lib.check(_lib.lib().cah_kmers_present_batch_host(f._plan.handle, 0, seqs.ctypes.data,
                                                           offsets.ctypes.data, len(c["reads"]),
                                                           present.ctypes.data))
        assert present.tolist() == [int(w) for _, w in c["reads"]], c["sets"]
    f = KmerFinder([(0, None, ["ACGT"])])
    assert f.kmers_present("ttacgtaa") is True and f.kmers_present("") is False
    with pytest.raises(ValueError):
        KmerFinder([(0, None, ["A" * 65])])
    with pytest.raises(TypeError):
        KmerFinder([(0, None, [b"ACGT"])])


def test_kmer_finder_many_words_and_clamping(hip, orc):
    """more packed words than fit in LDS (global-memory table path) + positive stop beyond the
    read end (clamped; undefined behaviour in the reference)"""
    from cutadapt_amd._kmer_finder import KmerFinder
    rng = random.Random(5)
    sets = [(0, None, [rs(rng, 40, "ACGT") for _ in range(45)]), (-30, None, [rs(rng, 9, "ACGT")]),
            (0, 200, [rs(rng, 12, "ACGT")])]
    f = KmerFinder(sets)
    o = orc.KmerFinder(sets)
    assert f.number_of_searches == 47
    reads = [rs(rng, rng.randint(0, 150), "ACGT") for _ in range(200)]
    for i in range(0, 200, 3):
        km = rng.choice(rng.choice(sets)[2])
        p = rng.randint(0, len(reads[i]))
        reads[i] = reads[i][:p] + km + reads[i][p:]
    assert [f.kmers_present(r) for r in reads[:40]] == [o.kmers_present(r) for r in reads[:40]]
    seqs, offsets = orc.pack_reads(reads)
    want = o.kmers_present_batch(seqs, offsets)
    from cutadapt_amd import _lib
    got = np.zeros(len(reads), dtype=np.uint8)
    _lib.check(_lib.lib().cah_kmers_present_batch_host(f._plan.handle, 0, seqs.ctypes.data, offsets.ctypes.data,
                                                       len(reads), got.ctypes.data))
    assert np.array_equal(got, want) and want.sum() > 30


def test_adapters_that_tolerate_as_many_errors_as_they_have_characters(hip, orc):
    """Error rate 1.0 (-e 4 on a 4-character adapter): kmer_heuristic emits an EMPTY k-mer next to single characters.  The
    reference accepts it and never finds it (_kmer_finder.pyx:121-160), so its prefilter turns away reads without any adapter
    character -- all-N reads -- that the aligner alone would match.  Same here: the finder takes the sets as they are, the
    library searches them without the empty k-mer; against the oracle on the reference's own sets."""
    from cutadapt_amd import adapters as A
    from cutadapt_amd._kmer_finder import KmerFinder
    from cutadapt_amd.batch import ReadBatch
    rng = random.Random(77)
    fixed = ["", "A", "CCC", "NNNN", "ACGT", "TTTTG", "nnnn", "RRRR"]
    for sets in ([(0, None, ["", "A"])], [(0, None, ["A", ""])], [(-5, None, ["", "G", "N"])], [(0, None, [""])]):
        f, o = KmerFinder(sets, True, False), orc.KmerFinder(sets, True, False)
        assert f.positions_and_kmers == sets
        reads = fixed + [rs(rng, rng.randint(0, 20), "ACGTN") for _ in range(30)]
        assert [f.kmers_present(r) for r in reads] == [o.kmers_present(r) for r in reads], sets
    turned_away = 0
    for cls, seq, errs in ((A.BackAdapter, "ACGT", 4), (A.BackAdapter, "ACGTAC", 6), (A.FrontAdapter, "TTGCA", 5),
                           (A.SuffixAdapter, "NNNNG", 1), (A.AnywhereAdapter, "GATC", 4)):
        ad = cls(seq, max_errors=errs, min_overlap=3)
        assert isinstance(ad.kmer_finder, KmerFinder) and ad.max_error_rate == 1.0
        assert any("" in k for _, _, k in ad.kmer_finder.positions_and_kmers)
        reads = ["NNNNNNNN", "nnnn", "", "N", "TTTTTTTT", "ACGTACGT", "GGGGACGG", "RYRYRYRY", "NNNNANNN"] + \
                [rs(rng, rng.randint(0, 30), "ACGTN") for _ in range(60)]
        spec = ad.matcher_spec()
        oa = orc.Aligner(spec.sequence, spec.max_error_rate, spec.flags, spec.wildcard_ref, spec.wildcard_query,
                         spec.indel_cost, spec.min_overlap)
        of = orc.KmerFinder(ad.kmer_finder.positions_and_kmers, ad.kmer_finder.ref_wildcards, ad.kmer_finder.query_wildcards)
        seqs, offsets = orc.pack_reads(reads)
        o6, st = orc.match_batch(oa, of, seqs, offsets)
        alone6, alone = orc.match_batch(oa, None, seqs, offsets)              # the aligner without a prefilter
        turned_away += int(((alone == 1) & (st != 1)).sum())
        bm = ad.match_to_batch(ReadBatch.from_strings(reads))
        assert np.array_equal(bm.found, st == 1), (cls.__name__, seq)
        assert np.array_equal(bm.coords[bm.found], o6[st == 1].astype(np.int64)), (cls.__name__, seq)
        for i in (0, 4, 5, 8):                                                # ... and read by read
            m = ad.match_to(reads[i])
            assert (m is not None) == bool(st[i] == 1) and (m is None or m.astuple() == tuple(int(x) for x in o6[i])), (seq, i)
    assert turned_away > 0                                    # the case exists: the prefilter decides, not the aligner


# ---------------------------------------------------------------------------------------------
# fused match path (prefilter -> queue -> DP) and adapters API
# ---------------------------------------------------------------------------------------------
def test_golden_illumina_info(hip, golden):
    """reference tests/cut/illumina.info.txt: errors, rstart, rstop of every read"""
    from cutadapt_amd.adapters import BackAdapter, RemoveAfterMatch
    g = golden("illumina_info.json")
    adapter = BackAdapter(g["adapter"], max_errors=g["max_errors"], min_overlap=g["min_overlap"])
    for s, want in list(zip(g["reads"], g["expected"]))[:12]:
        mt = adapter.match_to(s)
        got = None if mt is None else [mt.errors, mt.rstart, mt.rstop]
        assert got == want
        assert mt is None or isinstance(mt, RemoveAfterMatch)
    from cutadapt_amd.batch import ReadBatch
    bm = adapter.match_to_batch(ReadBatch.from_strings(g["reads"]))
    got = [[int(c[5]), int(c[2]), int(c[3])] if f else None for c, f in zip(bm.coords, bm.found)]
    assert got == g["expected"]
    assert sum(f for f in bm.found) == 56
    m0 = bm.match(0)
    assert m0.trimmed(g["reads"][0]) == g["reads"][0][:m0.rstart]


def test_golden_adapter_classes(hip, golden):
    """match_to of every adapter class vs reference (single-read API) and batch API"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    n = 0
    for c in golden("adapters.json"):
        adapter = getattr(A, c["cls"])(c["sequence"], **c["kwargs"])
        reads = [r for r, _ in c["reads"]]
        bm = adapter.match_to_batch(ReadBatch.from_strings(reads))
        for i, (read, want) in enumerate(c["reads"]):
            mt = bm.match(i)
            got = None if mt is None else {"cls": type(mt).__name__, "t": list(mt.astuple())}
            assert got == want, (c["cls"], c["sequence"], c["kwargs"], read, got, want)
            n += 1
        if n % 7 == 0:   # the per-read API on a subset (each call is a kernel launch)
            read, want = c["reads"][0]
            mt = adapter.match_to(read)
            got = None if mt is None else {"cls": type(mt).__name__, "t": list(mt.astuple())}
            assert got == want
    assert n >= 2000
    # the edges of the parameter space (tests/golden/make_adapters_extreme_golden.py: adapters of 1 .. 33 characters, error
    # rates up to 1.0, min_overlap beyond the adapter, force_anywhere on every class that takes it) through the batch API
    # -- ALL of them on the GPU (round-5 review: the force_anywhere slip of round 5 lived through that round's GPU packages
    # because only the CPU suite replayed this golden in full): force_anywhere on every class that takes it, rate 1.0
    n, kinds, full_rate = 0, set(), 0
    for c in golden("adapters_extreme.json"):
        adapter = getattr(A, c["cls"])(c["sequence"], **c["kwargs"])
        bm = adapter.match_to_batch(ReadBatch.from_strings([r for r, _ in c["reads"]]))
        kinds.add((c["cls"], bool(c["kwargs"].get("force_anywhere"))))
        full_rate += c["kwargs"].get("max_errors") == 1
        for i, (read, want) in enumerate(c["reads"]):
            mt = bm.match(i)
            got = None if mt is None else {"cls": type(mt).__name__, "t": list(mt.astuple())}
            assert got == want, (c["cls"], c["sequence"], c["kwargs"], read, got, want)
            n += 1
    assert n >= 3000
    assert sum(1 for k in kinds if k[1]) == 8 and full_rate >= 8, (kinds, full_rate)


def test_golden_linked_and_multiple(hip, golden):
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    g = golden("linked_multiple.json")

    def mj(m):
        return None if m is None else {"cls": type(m).__name__, "t": list(m.astuple())}

    for c in g["linked"]:
        front = getattr(A, c["front_cls"])(c["front"], max_errors=0.1)
        back = A.BackAdapter(c["back"], max_errors=0.1, min_overlap=3)
        linked = A.LinkedAdapter(front, back, front_required=c["front_required"],
                                 back_required=c["back_required"], name="linked")
        reads = [r for r, _ in c["reads"]]
        lb = linked.match_to_batch(ReadBatch.from_strings(reads))
        for i, (read, want) in enumerate(c["reads"]):
            mt = lb.match(i)
            got = None if mt is None else {"front": mj(mt.front_match), "back": mj(mt.back_match)}
            assert got == want, (c, read, got)
        read, want = c["reads"][0]
        mt = linked.match_to(read)
        got = None if mt is None else {"front": mj(mt.front_match), "back": mj(mt.back_match)}
        assert got == want
    for c in g["multiple"]:
        ads = [A.BackAdapter(s, max_errors=0.15, min_overlap=3) for s in c["seqs"]]
        multi = A.MultipleAdapters(ads)
        reads = [r for r, _ in c["reads"]]
        bm = multi.match_to_batch(ReadBatch.from_strings(reads))
        for i, (read, want) in enumerate(c["reads"]):
            mt = bm.match(i)
            got = None if mt is None else {"adapter": ads.index(mt.adapter), "m": mj(mt)}
            assert got == want, (c["seqs"], read, got, want)
        read, want = c["reads"][1]
        mt = multi.match_to(read)
        got = None if mt is None else {"adapter": ads.index(mt.adapter), "m": mj(mt)}
        assert got == want


def test_match_batch_vs_oracle_truseq_synthetic(hip, orc):
    """C1-sized workload (10k x 150 bp, TruSeq 3' adapter, e=0.1, O=3) generated on the GPU,
    matched through the fused path with device-resident buffers, compared with the oracle on
    the byte-identical CPU-generated reads."""
    import torch
    from cutadapt_amd.adapters import BackAdapter
    from cutadapt_amd.batch import ReadBatch, match_batch
    n, L = 10000, 150
    batch = ReadBatch.synthetic(n, L, [TRUSEQ], seed=1)
    seqs_cpu, offsets_cpu = orc.synth_reads(1, 0, n, L, [TRUSEQ])
    assert np.array_equal(batch.seqs.cpu().numpy(), seqs_cpu)          # generator twins agree
    assert np.array_equal(batch.offsets.cpu().numpy(), offsets_cpu)
    adapter = BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
    res = match_batch(adapter._fused_plan, batch)
    torch.cuda.synchronize()
    out6, status, best = res.cpu()
    oa = orc.Aligner(TRUSEQ, 0.1, BACK, False, False, 1, 3)
    of = orc.KmerFinder(adapter.kmer_finder.positions_and_kmers)
    want6, want_st = orc.match_batch(oa, of, seqs_cpu, offsets_cpu)
    assert_same(out6, status, want6, want_st, "fused match")
    assert 2000 < int(want_st.sum()) < 3500
    assert np.array_equal(best, np.where(want_st == 1, 0, -1))
    # the prefilter must not change results: locate alone gives the same matches
    res2 = adapter.aligner.locate_batch(batch)
    out6b, statusb, _ = res2.cpu()
    want6b, want_stb = oa.locate_batch(seqs_cpu, offsets_cpu)
    assert_same(out6b, statusb, want6b, want_stb, "locate_batch")


def test_large_batch_properties(hip, orc):
    """2M reads (size-independent properties + sampled oracle check): idempotence, shard
    independence (matching a sub-range alone gives the same rows), coordinate invariants."""
    import torch
    from cutadapt_amd.adapters import BackAdapter
    from cutadapt_amd.batch import ReadBatch, match_batch
    n, L = 2_000_000, 150
    adapter = BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
    batch = ReadBatch.synthetic(n, L, [TRUSEQ], seed=2)
    r1 = match_batch(adapter._fused_plan, batch)
    r2 = match_batch(adapter._fused_plan, batch)
    torch.cuda.synchronize()
    assert torch.equal(r1.out6, r2.out6) and torch.equal(r1.status, r2.status)      # idempotent
    out6, status = r1.out6, r1.status
    found = status == 1
    assert int((status == 2).sum()) == 0
    frac = float(found.float().mean())
    assert 0.2 < frac < 0.35, frac
    o = out6[found].to(torch.int64)
    # invariants of a 3' match: starts at adapter position 0, stops inside the read, one of
    # ref_start / query_start is zero, errors within the rate, score consistent with errors
    assert bool((o[:, 0] == 0).all()) and bool((o[:, 1] <= 33).all()) and bool((o[:, 1] >= 3).all())
    assert bool((o[:, 3] <= L).all()) and bool((o[:, 2] < o[:, 3]).all())
    assert bool((o[:, 5] <= (o[:, 1] * 0.1).floor()).all())
    assert bool((o[:, 4] <= o[:, 1]).all()) and bool((o[:, 4] >= o[:, 1] - 3 * o[:, 5]).all())
    # non-matching rows are zero
    assert int(out6[~found].abs().sum()) == 0
    # shard independence: rows [a, b) computed alone equal the same rows of the full run
    a, b = 777_777, 777_777 + 50_000
    sub = ReadBatch.synthetic(b - a, L, [TRUSEQ], seed=2, first_index=a)
    rs_ = match_batch(adapter._fused_plan, sub)
    torch.cuda.synchronize()
    assert torch.equal(rs_.out6, out6[a:b]) and torch.equal(rs_.status, status[a:b])
    # sampled oracle check on the same rows
    seqs_cpu, offsets_cpu = orc.synth_reads(2, a, b - a, L, [TRUSEQ])
    oa = orc.Aligner(TRUSEQ, 0.1, BACK, False, False, 1, 3)
    of = orc.KmerFinder(adapter.kmer_finder.positions_and_kmers)
    want6, want_st = orc.match_batch(oa, of, seqs_cpu, offsets_cpu)
    assert_same(rs_.out6.cpu().numpy(), rs_.status.cpu().numpy(), want6, want_st, "shard vs oracle")


def test_multi_adapter_plan_vs_oracle(hip, orc):
    """several adapters of mixed kinds in one plan: best-match rule on the device"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    rng = random.Random(31)
    seqs = [rs(rng, rng.randint(12, 40), "ACGT") for _ in range(7)]
    ads = [A.BackAdapter(s, max_errors=0.12, min_overlap=4) for s in seqs[:5]]
    ads.append(A.AnywhereAdapter(seqs[5], max_errors=0.1))
    ads.append(A.SuffixAdapter(seqs[6], max_errors=0.2, indels=False))
    multi = A.MultipleAdapters(ads)
    reads = []
    for _ in range(3000):
        r = rs(rng, rng.randint(20, 120), "ACGT")
        if rng.random() < 0.7:
            s = rng.choice(seqs)
            p = rng.randint(0, len(r))
            r = r[:p] + s[:rng.randint(4, len(s))] + (r[p:] if rng.random() < 0.5 else "")
        reads.append(r)
    bm = multi.match_to_batch(ReadBatch.from_strings(reads))
    # oracle: run every adapter separately, apply the reference's argmax rule
    oseqs, ooffs = orc.pack_reads(reads)
    best = [None] * len(reads)
    for idx, ad in enumerate(ads):
        spec = ad.matcher_spec()
        if spec.kind == 0:
            oa = orc.Aligner(spec.sequence, spec.max_error_rate, spec.flags, spec.wildcard_ref,
                             spec.wildcard_query, spec.indel_cost, spec.min_overlap)
            of = orc.KmerFinder(spec.kmer_sets, spec.kmer_ref_wildcards, spec.kmer_query_wildcards) \
                if spec.kmer_sets is not None else None
            o6, st = orc.match_batch(oa, of, oseqs, ooffs)
        else:
            oc = orc.SuffixComparer(spec.sequence, spec.max_error_rate, spec.wildcard_ref,
                                    spec.wildcard_query, spec.min_overlap)
            o6, st = oc.locate_batch(oseqs, ooffs)
        for i in range(len(reads)):
            if st[i] != 1:
                continue
            t = tuple(int(v) for v in o6[i])
            if best[i] is None or t[4] > best[i][1][4] or (t[4] == best[i][1][4] and t[5] < best[i][1][5]):
                best[i] = (idx, t)
    for i in range(len(reads)):
        got = (int(bm.adapter_index[i]), tuple(int(v) for v in bm.coords[i])) if bm.found[i] else None
        assert got == best[i], (reads[i], got, best[i])
    assert sum(b is not None for b in best) > 1000


def test_column_skipping_is_exact(hip, orc):
    """3' adapters whose prefilter has the pigeonhole property let the DP start shortly before
    the first k-mer hit (DESIGN.md "Column skipping").  The fused path must stay bit-identical to
    the oracle, which always computes every column: random adapters / rates / indel costs /
    wildcard modes, reads with several (partial, edited, repeated) adapter copies at all
    positions, long reads (keys are clipped at 63 chunks)."""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    rng = random.Random(777)
    total = hits = 0
    for it in range(70):
        m = rng.choice([8, 12, 17, 20, 25, 30, 33, 34, 40, 48, 57, 64])
        al = "ACGT" if rng.random() < 0.7 else "ACGTN"
        seq = rs(rng, m, al)
        if rng.random() < 0.15:
            seq = (seq[:max(2, m // 4)] * 5)[:m]                  # repetitive adapter
        if set(seq) == {"N"}:
            seq = "A" + seq[1:]                                   # (an adapter of N only is rejected, like the reference does)
        kwargs = {"max_errors": rng.choice([0, 0.05, 0.1, 0.1, 0.15, 0.2, 0.3]),
                  "min_overlap": rng.randint(1, 6), "read_wildcards": rng.random() < 0.25,
                  "indels": rng.random() < 0.8}
        ad = A.BackAdapter(seq, **kwargs)
        reads = []
        for _ in range(1500):
            n = rng.choice([rng.randint(0, 40), rng.randint(100, 200), 150, 150, rng.randint(900, 1300)])
            r = list(rs(rng, n, "ACGT"))
            for _copy in range(rng.choice([0, 1, 1, 1, 2, 3])):
                a0 = rng.randint(0, m - 1) if rng.random() < 0.2 else 0
                piece = list(ad.sequence[a0:a0 + rng.randint(1, m)])
                for _e in range(rng.choice([0, 0, 0, 1, 1, 2, 3])):
                    if piece:
                        x = rng.randrange(len(piece))
                        op = rng.randint(0, 2)
                        if op == 0:
                            piece[x] = rng.choice("ACGT")
                        elif op == 1:
                            piece.insert(x, rng.choice("ACGT"))
                        else:
                            del piece[x]
                p0 = rng.randint(0, len(r))
                r[p0:p0] = piece
            if rng.random() < 0.2:
                r = [c if rng.random() > 0.02 else "N" for c in r]
            reads.append("".join(r)[:rng.choice([len(r), 150, len(r)])])
        bm = ad.match_to_batch(ReadBatch.from_strings(reads))
        spec = ad.matcher_spec()
        oa = orc.Aligner(spec.sequence, spec.max_error_rate, spec.flags, spec.wildcard_ref, spec.wildcard_query,
                         spec.indel_cost, spec.min_overlap)
        of = orc.KmerFinder(spec.kmer_sets, spec.kmer_ref_wildcards, spec.kmer_query_wildcards)
        seqs, offsets = orc.pack_reads(reads)
        want6, want_st = orc.match_batch(oa, of, seqs, offsets)
        found = want_st == 1
        assert np.array_equal(bm.found, found), (seq, kwargs, np.nonzero(bm.found != found)[0][:5])
        bad = np.nonzero((bm.coords != want6.astype(np.int64)).any(axis=1) & found)[0]
        assert len(bad) == 0, (seq, kwargs, reads[bad[0]], bm.coords[bad[0]], want6[bad[0]])
        total += len(reads)
        hits += int(found.sum())
    # (drawn cases: the share of reads with a hit moves with the seed -- looser under shifted seeds, where it fell to 29 k)
    assert total >= 100000 and hits > (30000 if not os.environ.get("CAH_TEST_SEED_OFFSET") else 20000)


def test_column_skipping_equals_full_dp_large(hip):
    """GPU vs GPU at scale: fused path (prefilter -> ordered queue -> DP that may skip columns)
    against kmers_present_batch AND locate_batch (full DP over every column, itself checked
    against the oracle above) on ~6 M synthetic reads over random 3' adapters."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(2718)
    n = 150_000
    checked = 0
    for it in range(40):
        m = rng.choice([10, 16, 21, 28, 33, 33, 36, 45, 52, 64])
        seq = rs(rng, m, "ACGT")
        if it % 7 == 0:
            seq = (seq[:5] * 20)[:m]
        ad = A.BackAdapter(seq, max_errors=rng.choice([0.05, 0.1, 0.1, 0.2, 0.25]), min_overlap=rng.randint(1, 5),
                           indels=rng.random() < 0.85)
        L = rng.choice([50, 100, 150, 150, 251])
        batch = ReadBatch.synthetic(n, L, [ad.sequence], seed=1000 + it, p_adapter=rng.choice([0.3, 0.6, 0.9]),
                                    p_edit=rng.choice([0.0, 0.02, 0.05, 0.1]), p_n=rng.choice([0.0, 0.005, 0.02]))
        fused = match_batch(ad._fused_plan, batch)
        present = ad.kmer_finder.kmers_present_batch(batch)
        full = ad.aligner.locate_batch(batch)
        torch.cuda.synchronize()
        want_found = (present == 1) & (full.status == 1)
        assert torch.equal(fused.status == 1, want_found), (seq, it)
        assert torch.equal(fused.out6[want_found], full.out6[want_found]), (seq, it)
        assert int(fused.out6[~want_found].abs().sum()) == 0
        checked += int(want_found.sum())
    assert checked > 1_000_000


def test_edge_cases_through_the_batch_api(hip, orc):
    """empty batch, empty reads, a very long read, reads beyond the length limit, views,
    ASCII validation, workspace too small"""
    import torch
    from cutadapt_amd import _lib
    from cutadapt_amd.adapters import BackAdapter
    from cutadapt_amd.batch import ReadBatch, match_batch, locate_batch
    ad = BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
    # empty batch
    empty = ReadBatch.from_strings([])
    r = match_batch(ad._fused_plan, empty)
    assert r.status.numel() == 0 and r.out6.shape == (0, 6)
    assert ad.match_to_batch(empty).tuples() == []
    # only empty reads / one-character reads
    b = ReadBatch.from_strings(["", "", "A", ""])
    assert ad.match_to_batch(b).tuples() == [None, None, None, None]
    # a long read with the adapter far inside, next to short ones (ragged batch)
    rng = random.Random(9)
    long_read = rs(rng, 200_000, "ACGT") + TRUSEQ + rs(rng, 1000, "ACGT")
    reads = ["ACGT" * 10 + TRUSEQ[:12], long_read, "", TRUSEQ, rs(rng, 5000, "ACGT")]
    got = ad.match_to_batch(ReadBatch.from_strings(reads)).tuples()
    oa = orc.Aligner(TRUSEQ, 0.1, BACK, False, False, 1, 3)
    of = orc.KmerFinder(ad.kmer_finder.positions_and_kmers)
    want = [oa.locate(x) if of.kmers_present(x) else None for x in reads]
    assert got == want and got[1][2] == 200_000 and got[3] == (0, 33, 0, 33, 33, 0)
    # beyond CAH_MAX_READ_LEN: flagged invalid, not silently mishandled
    too_long = ReadBatch.from_strings(["ACGT", "A" * 1_000_001])
    res = locate_batch(ad.aligner._plan, 0, too_long)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist()[1] == 2
    with pytest.raises(ValueError):
        ad.match_to_batch(too_long)
    # sub-sequence views (second stage of linked adapters): same result as slicing on the host
    base = ReadBatch.from_strings(reads)
    starts = torch.tensor([3, 150_000, 0, 1, 10], device=base.device)
    lens = base.lengths() - starts
    got_v = ad.match_to_batch(base.view(starts, lens)).tuples()
    want_v = [oa.locate(x[s:]) if of.kmers_present(x[s:]) else None for x, s in zip(reads, starts.cpu().tolist())]
    assert got_v == want_v
    # ASCII validation of externally supplied buffers
    raw = np.frombuffer(b"ACGT" + b"AC\xffT" + b"ACGT", dtype=np.uint8)
    bad = ReadBatch.from_host(raw, np.array([0, 4, 8, 12], dtype=np.int64))
    with pytest.raises(ValueError):
        bad.validate_ascii()
    ReadBatch.from_host(raw[:4], np.array([0, 4], dtype=np.int64)).validate_ascii()
    # workspace too small is an error, not a crash
    n = 1000
    bb = ReadBatch.synthetic(n, 150, [TRUSEQ], seed=3)
    out6 = torch.empty((n, 6), dtype=torch.int32, device=bb.device)
    st = torch.empty(n, dtype=torch.uint8, device=bb.device)
    small = torch.empty(64, dtype=torch.uint8, device=bb.device)
    rc = _lib.lib().cah_match_batch(ad._fused_plan.handle, bb.seqs.data_ptr(), bb.offsets.data_ptr(), None, n,
                                    out6.data_ptr(), None, st.data_ptr(), small.data_ptr(), small.numel(), None)
    assert rc == _lib.CAH_EINVAL and "workspace" in _lib.last_error()


@pytest.mark.gpu
def test_lean_prefilter_equals_general_prefilter(hip, orc):
    """Equal-length batches of 3' adapters go through k_filter_lean<UNIFORM>, the same reads passed as a
    view (explicit lens) through its ragged variant: identical results, and both equal the oracle
    (plans that are not eligible -- 5' / anywhere adapters, wide k-mers -- keep using k_filter and are
    covered by the other tests).  Read lengths around every boundary of the tail windows and chunks."""
    import random
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(77)
    for it in range(14):
        m = rng.choice([8, 13, 20, 33, 33, 40, 57, 64])
        seq = rs(rng, m, "ACGT") if it % 4 else rs(rng, m, "ACGTN")
        ad = A.BackAdapter(seq, max_errors=rng.choice([0.0, 0.1, 0.1, 0.2]), min_overlap=rng.randint(1, 6),
                           read_wildcards=rng.random() < 0.2, indels=rng.random() < 0.8)
        for n in sorted({0, 1, 2, 5, 15, 16, 17, rng.randint(18, 31), 32, 33, 47, 48, 49, 64, rng.randint(65, 140), 150}):
            reads = []
            for _ in range(300):
                r = list(rs(rng, n, "ACGT"))
                if n and rng.random() < 0.6:
                    p = rng.randint(0, n - 1)
                    piece = list(ad.sequence[:rng.randint(1, m)])
                    for _e in range(rng.choice([0, 0, 1, 2])):
                        if piece:
                            piece[rng.randrange(len(piece))] = rng.choice("ACGT")
                    r[p:p + len(piece)] = piece
                    r = r[:n]
                if n and rng.random() < 0.1:
                    r[rng.randrange(n)] = "N"
                reads.append("".join(r))
            batch = ReadBatch.from_strings(reads)
            lean = match_batch(ad._fused_plan, batch).cpu()
            lens = torch.full((len(reads),), n, dtype=torch.int32, device=batch.device)
            view = ReadBatch(batch.seqs, batch.offsets[:len(reads)].clone(), lens, n_reads=len(reads), validated=True)
            general = match_batch(ad._fused_plan, view).cpu()
            assert np.array_equal(lean[1], general[1]) and np.array_equal(lean[0], general[0]), (seq, n)
            finder = orc.KmerFinder(ad.kmer_finder.positions_and_kmers, ad.adapter_wildcards, ad.read_wildcards) \
                if hasattr(ad.kmer_finder, "positions_and_kmers") else None
            al = ad.aligner
            oal = orc.Aligner(ad.sequence, ad.max_error_rate, flags=14, wildcard_ref=ad.adapter_wildcards,
                              wildcard_query=ad.read_wildcards, indel_cost=1 if ad.indels else 100000,
                              min_overlap=ad.min_overlap)
            for i, r in enumerate(reads):
                exp = oal.locate(r) if (finder is None or finder.kmers_present(r)) else None
                got = tuple(int(v) for v in lean[0][i]) if lean[1][i] == 1 else None
                assert got == exp, (seq, n, r, got, exp)


def test_anchored_adapters_without_errors_vs_oracle(hip, orc):
    """k_anchored_exact: anchored aligners (Where.PREFIX = 8, Where.SUFFIX = 2) whose threshold at full length is 0 --
    short adapters, adapters of mostly N wildcards, rate 0 -- take a character comparison instead of the cell DP.
    Reads that start / end with the adapter (exact, one edit, shifted by one, truncated), wildcards on either
    side, every indel cost, invalid bytes inside and outside the columns the aligner reads."""
    from cutadapt_amd import _lib
    rng = random.Random(808)
    total = found = 0
    for it in range(120):
        flags = rng.choice([PREFIX, SUFFIX])
        wr = rng.random() < 0.5
        m = rng.choice([1, 2, 3, 5, 8, 9, 12, 16, 20, 33, 40, 64])
        if wr and rng.random() < 0.6:
            adapter = "".join(rng.choice("ACGTNNNN") for _ in range(m))
            if set(adapter) <= set("N"):
                adapter = "A" + adapter[1:]
        else:
            adapter = rs(rng, m, "ACGT")
        eff = m - (adapter.count("N") if wr else 0)
        rate = rng.choice([0.0, 0.05, 0.1, 0.99 / max(eff, 1)])
        args = (adapter, rate, flags, wr, rng.random() < 0.3, rng.choice([1, 1, 2, 100000]), rng.randint(1, m))
        oa = orc.Aligner(*args)
        plan = _lib.Plan([_lib.MatcherSpec(*args)])
        reads = []
        for _ in range(400):
            body = rs(rng, rng.randint(0, 40), "ACGT")
            piece = list(adapter.replace("N", rng.choice("ACGT")) if wr else adapter)
            u = rng.random()
            if u < 0.35 and piece:
                x = rng.randrange(len(piece))
                op = rng.randint(0, 2)
                if op == 0:
                    piece[x] = rng.choice("ACGTN")
                elif op == 1:
                    piece.insert(x, rng.choice("ACGT"))
                else:
                    del piece[x]
            elif u < 0.45:
                piece = piece[:rng.randint(0, len(piece))]
            p = "".join(piece)
            if rng.random() < 0.15:
                p = p.lower()
            q = p + body if flags == PREFIX else body + p
            if rng.random() < 0.1:
                q = rng.choice("ACGT") + q if flags == PREFIX else q + rng.choice("ACGT")
            reads.append(q)
        seqs, offsets = orc.pack_reads(reads)
        seqs = seqs.copy()
        k = int(rate * m)
        for _ in range(6):                                   # bytes >= 0x80 inside the columns the aligner reads
            r = rng.randrange(len(reads))                    # (the first / last min(n, m + k): _align.pyx:346-352)
            n = int(offsets[r + 1] - offsets[r])
            span = min(n, m + k)
            if span:
                lo = offsets[r] if flags == PREFIX else offsets[r + 1] - span
                seqs[rng.randrange(lo, lo + span)] = 0xC3
        want6, want_st = oa.locate_batch(seqs, offsets)
        got6, got_st = host_locate_batch(plan, seqs, offsets)
        assert_same(got6, got_st, want6, want_st, f"anchored {args}")
        total += len(reads)
        found += int((want_st == 1).sum())
    assert total > 40000 and found > 8000, (total, found)


def test_reverse_reads_kernel(hip):
    """cah_reverse_reads_batch (what Rightmost* adapters search, reference adapters.py:766, :870): every length around
    the 16-character chunks, empty reads, a view (explicit lengths into a longer buffer), a large ragged batch."""
    import torch
    from cutadapt_amd.adapters import _reverse_batch
    from cutadapt_amd.batch import ReadBatch
    rng = random.Random(31)
    reads = [rs(rng, n, "ACGTN") for n in list(range(0, 70)) + [100, 150, 151, 255, 256, 257, 1000]]
    reads += [rs(rng, rng.randint(0, 300), "ACGTacgtN") for _ in range(20000)]
    batch = ReadBatch.from_strings(reads)
    rev = _reverse_batch(batch)
    seqs = rev.seqs.cpu().numpy().tobytes().decode()
    offs = rev.offsets.cpu().numpy()
    assert len(offs) == len(reads) + 1 and offs[-1] == sum(len(r) for r in reads)
    for i, r in enumerate(reads):
        assert seqs[offs[i]:offs[i + 1]] == r[::-1], (i, len(r))
    # a view: the middle part of every read
    lens = batch.lengths()
    starts = torch.clamp(lens // 3, max=40).to(torch.int32)
    view = batch.view(starts, lens - 2 * starts.to(torch.int64))
    rv = _reverse_batch(view)
    seqs = rv.seqs.cpu().numpy().tobytes().decode()
    offs = rv.offsets.cpu().numpy()
    for i, r in enumerate(reads[:3000]):
        s0 = min(len(r) // 3, 40)
        assert seqs[offs[i]:offs[i + 1]] == r[s0:len(r) - s0][::-1], (i, len(r))
