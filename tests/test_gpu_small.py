"""Batches of one to a few reads through the host-pointer entry points -- what the per-read match_to() / locate() /
kmers_present() calls of the mirror classes send; they take the tiny-batch route of the library (one memset for all
counters, no batch check, contiguous outputs cleared at once) -- against the oracle.  GPU only."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rs(rng, n, al="ACGT"):
    return "".join(rng.choice(al) for _ in range(n))


def _host(fn, plan, seqs, offsets, adapter=None):
    from cutadapt_amd import _lib
    n = len(offsets) - 1
    out6 = np.zeros((n, 6), dtype=np.int32)
    status = np.zeros(n, dtype=np.uint8)
    best = np.zeros(n, dtype=np.int32)
    L = _lib.lib()
    if fn == "locate":
        _lib.check(L.cah_locate_batch_host(plan.handle, adapter, seqs.ctypes.data, offsets.ctypes.data, n, out6.ctypes.data, status.ctypes.data))
    elif fn == "present":
        _lib.check(L.cah_kmers_present_batch_host(plan.handle, adapter, seqs.ctypes.data, offsets.ctypes.data, n, status.ctypes.data))
    else:
        _lib.check(L.cah_match_batch_host(plan.handle, seqs.ctypes.data, offsets.ctypes.data, n, out6.ctypes.data, best.ctypes.data, status.ctypes.data))
    return out6, status, best


def test_small_batches_vs_oracle_and_batch_kernels(hip, orc):
    from cutadapt_amd import _lib
    from cutadapt_amd import adapters as A
    rng = random.Random(1601)
    checked = 0
    for it in range(150):
        m = rng.choice([1, 4, 12, 33, 64, 65, 120])
        flags = rng.randint(0, 15)
        wr, wq = rng.random() < 0.3, rng.random() < 0.3
        adapter = rs(rng, m, "ACGT" * 4 + ("NRY" if wr else ""))
        rate = rng.choice([0.0, 0.1, 0.2, 0.4])
        D = rng.choice([1, 1, 2, 100000])
        O = rng.choice([1, 3])
        reads = []
        for _ in range(rng.randint(1, 16)):
            r = rs(rng, rng.randint(0, 180), "ACGTN" if wq else "ACGT")
            if rng.random() < 0.7 and m:
                piece = "".join(c if c in "ACGT" else rng.choice("ACGT") for c in adapter)
                cut = rng.randint(0, len(piece))
                piece = piece[cut:] if rng.random() < 0.5 else piece[:max(1, cut)]
                pos = rng.randint(0, len(r))
                r = r[:pos] + piece + r[pos:]
            reads.append(r)
        seqs, offsets = orc.pack_reads(reads)
        # Aligner.locate
        spec = _lib.MatcherSpec(adapter, rate, flags, wr, wq, D, O)
        try:
            oa = orc.Aligner(adapter, rate, flags, wr, wq, D, O)
        except ValueError:
            continue
        plan = _lib.Plan([spec])
        want6, want_st = oa.locate_batch(seqs, offsets)
        got6, got_st, _ = _host("locate", plan, seqs, offsets, 0)
        assert np.array_equal(got_st, want_st) and np.array_equal(got6, want6), ("locate", it, adapter, flags, reads)
        # comparers
        if 0.0 <= rate <= 1.0:
            for kind, cls in ((_lib.KIND_PREFIX, orc.PrefixComparer), (_lib.KIND_SUFFIX, orc.SuffixComparer)):
                oc = cls(adapter, rate, wr, wq, O)
                w6, wst = oc.locate_batch(seqs, offsets)
                g6, gst, _ = _host("locate", _lib.Plan([_lib.MatcherSpec(adapter, rate, 0, wr, wq, 1, O, kind=kind)]), seqs, offsets, 0)
                assert np.array_equal(gst, wst) and np.array_equal(g6, w6), ("comparer", it, kind)
        checked += len(reads)
    # adapter classes: fused prefilter + aligner, several adapters, per-read API == batch kernels == oracle rule
    ads = [A.BackAdapter(rs(rng, 30), max_errors=0.1, min_overlap=3), A.FrontAdapter(rs(rng, 25), max_errors=0.15, min_overlap=4),
           A.AnywhereAdapter(rs(rng, 20), max_errors=0.1), A.BackAdapter(rs(rng, 90), max_errors=0.1, min_overlap=3)]
    plan = _lib.Plan([a.matcher_spec() for a in ads])
    for it in range(60):
        reads = []
        for _ in range(rng.randint(1, 16)):
            r = rs(rng, rng.randint(0, 150))
            if rng.random() < 0.8:
                a = rng.choice(ads).sequence
                r = r + a[:rng.randint(3, len(a))] if rng.random() < 0.5 else a[-rng.randint(3, len(a)):] + r
            reads.append(r)
        seqs, offsets = orc.pack_reads(reads)
        g6, gst, gb = _host("match", plan, seqs, offsets)
        want6 = np.zeros((len(reads), 6), dtype=np.int32); wst = np.zeros(len(reads), dtype=np.uint8); wb = np.full(len(reads), -1)
        for i, a in enumerate(ads):
            sp = a.matcher_spec()
            oa = orc.Aligner(sp.sequence, sp.max_error_rate, sp.flags, sp.wildcard_ref, sp.wildcard_query, sp.indel_cost, sp.min_overlap)
            of = orc.KmerFinder(sp.kmer_sets, sp.kmer_ref_wildcards, sp.kmer_query_wildcards) if sp.kmer_sets is not None else None
            c6, st = orc.match_batch(oa, of, seqs, offsets)
            f = st == 1
            better = f & ((wst == 0) | (c6[:, 4] > want6[:, 4]) | ((c6[:, 4] == want6[:, 4]) & (c6[:, 5] < want6[:, 5])))
            want6[better] = c6[better]; wst[better] = 1; wb[better] = i
        assert np.array_equal(gst, wst) and np.array_equal(g6, want6), ("match", it, reads)
        assert np.array_equal(gb, wb), ("best", it, gb, wb)
        # the same adapter alone: the single-adapter tiny route (one header memset, ragged prefilter)
        one = _lib.Plan([ads[0].matcher_spec()])
        sp = ads[0].matcher_spec()
        oa = orc.Aligner(sp.sequence, sp.max_error_rate, sp.flags, sp.wildcard_ref, sp.wildcard_query, sp.indel_cost, sp.min_overlap)
        of = orc.KmerFinder(sp.kmer_sets, sp.kmer_ref_wildcards, sp.kmer_query_wildcards)
        c6, st = orc.match_batch(oa, of, seqs, offsets)
        o6, ost, ob = _host("match", one, seqs, offsets)
        assert np.array_equal(ost, st) and np.array_equal(o6, c6) and np.array_equal(ob, np.where(st == 1, 0, -1))
        for i, a in enumerate(ads):
            if a.kmer_finder.__class__.__name__ != "KmerFinder":
                continue
            of = orc.KmerFinder(a.kmer_finder.positions_and_kmers)
            want = of.kmers_present_batch(seqs, offsets)
            _, gp, _ = _host("present", plan, seqs, offsets, i)
            assert np.array_equal(gp, want), ("present", it, i)
        checked += len(reads)
    assert checked > 1000
    # invalid input is flagged by both paths
    seqs, offsets = orc.pack_reads([b"ACGT\xc3\xa9ACGT", b"ACGT"])
    _, st, _ = _host("match", plan, seqs, offsets)
    assert st[0] == 2 and st[1] != 2


def test_one_read_calls_vs_oracle(hip, orc):
    """Adapter.match_to(str) / Aligner.locate(str): the one-read entry points (k_tiny: prefilter + cost scan of the
    read in one single-wave launch, the cell DP only when the scan leaves the read to it) against the oracle and
    against the general path (CAH_NO_TINY=1), over 3' adapters of every slot class, reads of every class of the
    scan (none / exact / substitutions / indels / partial at the end), empty and non-ASCII reads."""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.align import Aligner
    rng = random.Random(909)
    checked = dp_like = 0
    for it in range(40):
        m = rng.choice([3, 8, 13, 20, 33, 33, 40, 57, 64])
        adapter = rs(rng, m)
        kw = dict(max_errors=rng.choice([0.0, 0.1, 0.1, 0.2]), min_overlap=rng.randint(1, 5),
                  read_wildcards=rng.random() < 0.2, indels=rng.random() < 0.85)
        ad = A.BackAdapter(adapter, **kw)
        spec = ad.matcher_spec()
        oa = orc.Aligner(spec.sequence, spec.max_error_rate, spec.flags, spec.wildcard_ref, spec.wildcard_query,
                         spec.indel_cost, spec.min_overlap)
        of = orc.KmerFinder(spec.kmer_sets, spec.kmer_ref_wildcards, spec.kmer_query_wildcards)
        al = Aligner(adapter, kw["max_errors"], flags=14, wildcard_query=kw["read_wildcards"],
                     indel_cost=1 if kw["indels"] else 100000, min_overlap=kw["min_overlap"])
        oal = orc.Aligner(adapter, kw["max_errors"], 14, False, kw["read_wildcards"], 1 if kw["indels"] else 100000,
                          kw["min_overlap"])
        for _ in range(60):
            n = rng.choice([0, 1, 5, 20, 75, 150, 150, 301])
            r = list(rs(rng, n, "ACGTN" if rng.random() < 0.2 else "ACGT"))
            if n and rng.random() < 0.8:
                piece = list(adapter[:rng.randint(1, m)] if rng.random() < 0.3 else adapter)
                for _e in range(rng.choice([0, 0, 1, 1, 2, 3])):
                    x = rng.randrange(len(piece))
                    op = rng.random()
                    if op < 0.6:
                        piece[x] = rng.choice("ACGT")
                    elif op < 0.8:
                        piece.insert(x, rng.choice("ACGT"))
                    elif len(piece) > 1:
                        del piece[x]
                pos = rng.randint(0, n)
                r[pos:pos + len(piece)] = piece
                r = r[:n]
            read = "".join(r)
            want = oa.locate(read) if of.kmers_present(read) else None
            got = ad.match_to(read)
            sig = None if got is None else (got.astart, got.astop, got.rstart, got.rstop, got.score, got.errors)
            assert sig == want, (adapter, kw, read, sig, want)
            os.environ["CAH_NO_TINY"] = "1"
            try:
                got2 = ad.match_to(read)
            finally:
                os.environ.pop("CAH_NO_TINY", None)
            sig2 = None if got2 is None else (got2.astart, got2.astop, got2.rstart, got2.rstop, got2.score, got2.errors)
            assert sig2 == want
            assert al.locate(read) == oal.locate(read), (adapter, kw, read)
            checked += 1
            dp_like += want is not None and want[5] > 0
    assert checked == 2400 and dp_like > (200 if not os.environ.get("CAH_TEST_SEED_OFFSET") else 100)   # (drawn cases: looser under shifted seeds)
    ad = A.BackAdapter("AGATCGGAAGAGC", max_errors=0.1, min_overlap=3)
    with pytest.raises(ValueError):
        ad.match_to("ACGTéACGT")
    with pytest.raises(TypeError):
        ad.match_to(b"ACGT")
