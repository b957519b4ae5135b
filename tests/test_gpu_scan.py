"""k_back_scan + windowed k_dp_packed (GPU): the cost scan in front of the cell DP must not change a
single tuple.  Compared with the oracle on fuzz batches and, at scale, with the plain cell-DP path of the
same library (plans built under CAH_NO_SCAN=1)."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"


def _plan(adapter, rate, min_overlap, wr=False, wq=False, kmers=True, scan=True):
    from cutadapt_amd import _lib
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    sets = None
    if kmers:
        sets = create_positions_and_kmers(adapter.upper(), min_overlap, rate, back_adapter=True, front_adapter=False)
    old = os.environ.get("CAH_NO_SCAN")
    try:
        if scan:
            os.environ.pop("CAH_NO_SCAN", None)
        else:
            os.environ["CAH_NO_SCAN"] = "1"
        return _lib.Plan([_lib.MatcherSpec(adapter, rate, 14, wr, wq, 1, min_overlap, kmer_sets=sets,
                                           kmer_ref_wildcards=wr, kmer_query_wildcards=wq)])
    finally:
        if old is None:
            os.environ.pop("CAH_NO_SCAN", None)
        else:
            os.environ["CAH_NO_SCAN"] = old


def _host_locate(plan, seqs, offsets):
    from cutadapt_amd import _lib
    n = len(offsets) - 1
    out6 = np.zeros((n, 6), dtype=np.int32)
    status = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.lib().cah_locate_batch_host(plan.handle, 0, seqs.ctypes.data, offsets.ctypes.data, n,
                                                out6.ctypes.data, status.ctypes.data))
    return out6, status


def _host_match(plan, seqs, offsets):
    from cutadapt_amd import _lib
    n = len(offsets) - 1
    out6 = np.zeros((n, 6), dtype=np.int32)
    status = np.zeros(n, dtype=np.uint8)
    best = np.zeros(n, dtype=np.int32)
    _lib.check(_lib.lib().cah_match_batch_host(plan.handle, seqs.ctypes.data, offsets.ctypes.data, n,
                                               out6.ctypes.data, best.ctypes.data, status.ctypes.data))
    return out6, status


def _same(a6, ast, b6, bst, what):
    assert np.array_equal(ast, bst), f"{what}: status differs at {np.nonzero(ast != bst)[0][:10]}"
    bad = np.nonzero((a6 != b6).any(axis=1))[0]
    assert len(bad) == 0, f"{what}: tuples differ at {bad[:10]}: {a6[bad[:3]]} vs {b6[bad[:3]]}"


def _reads(rng, adapter, n_reads, max_len, p_edit, p_n, alphabet="ACGT"):
    reads = []
    for _ in range(n_reads):
        n = rng.randint(0, max_len)
        s = [rng.choice(alphabet) for _ in range(n)]
        for _copy in range(rng.randint(0, 2)):
            ad = []
            for c in adapter:
                u = rng.random()
                if u < p_edit / 2:
                    ad.append(rng.choice(alphabet))
                elif u < p_edit * 0.75:
                    ad.append(rng.choice(alphabet)); ad.append(c)
                elif u < p_edit:
                    pass
                else:
                    ad.append(c)
            if rng.random() < 0.3:
                ad = ad[:rng.randint(0, len(ad))]
                pos = max(0, n - len(ad))
            else:
                pos = rng.randint(0, n)
            s[pos:pos + len(ad)] = ad
            s = s[:n]
        reads.append("".join("N" if rng.random() < p_n else c for c in s))
    return reads


def test_scan_fuzz_vs_oracle(hip, orc):
    """random 3' adapters (m 1..64, error rates, min_overlap, wildcards) through locate (scan over every
    read) and through the fused prefilter -> scan -> DP path"""
    rng = random.Random(777)
    total = 0
    for it in range(120):
        m = rng.randint(1, 64)
        wr = it % 5 == 3
        wq = it % 7 == 2
        letters = "ACGT" * 4 + ("NRYSWKMBDHV" if wr else "")
        adapter = "".join(rng.choice(letters) for _ in range(m))
        if wr and all(c == "N" for c in adapter):
            adapter = "A" + adapter[1:]
        rate = rng.choice([0.0, 0.05, 0.1, 0.15, 0.2, 0.3])
        min_overlap = rng.choice([1, 2, 3, 5, max(1, m)])
        reads = _reads(rng, adapter if not wr else "".join(c if c in "ACGT" else rng.choice("ACGT") for c in adapter),
                       600, rng.choice([10, 40, 100, 170]), rng.choice([0.0, 0.03, 0.1, 0.2]),
                       rng.choice([0.0, 0.01, 0.1]), alphabet="ACGTacgtN" if wq else "ACGT")
        seqs, offsets = orc.pack_reads(reads)
        oa = orc.Aligner(adapter, rate, 14, wr, wq, 1, min_overlap)
        want6, want_st = oa.locate_batch(seqs, offsets)
        plan = _plan(adapter, rate, min_overlap, wr, wq, kmers=False)
        got6, got_st = _host_locate(plan, seqs, offsets)
        _same(got6, got_st, want6, want_st, f"locate it {it} adapter {adapter} rate {rate} O {min_overlap} wr {wr} wq {wq}")
        if not wr and not wq and m >= 3:
            plan = _plan(adapter, rate, min_overlap)
            from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
            of = orc.KmerFinder(create_positions_and_kmers(adapter, min_overlap, rate, back_adapter=True, front_adapter=False))
            w6, wst = orc.match_batch(oa, of, seqs, offsets)
            g6, gst = _host_match(plan, seqs, offsets)
            _same(g6, gst, w6, wst, f"match it {it} adapter {adapter} rate {rate} O {min_overlap}")
        total += len(reads)
    assert total > 60_000


def test_scan_equals_plain_dp_at_scale(hip):
    """4 M synthetic reads per read model: fused path with the cost scan == fused path without it"""
    import torch
    from cutadapt_amd.batch import ReadBatch, match_batch
    for gen in (dict(p_adapter=0.25, p_edit=0.02, p_n=0.005), dict(p_adapter=1.0, p_edit=0.1, p_n=0.02),
                dict(p_adapter=0.5, p_edit=0.3, p_n=0.0)):
        batch = ReadBatch.synthetic(4_000_000, 150, [TRUSEQ], seed=31, **gen)
        a = match_batch(_plan(TRUSEQ, 0.1, 3, scan=True), batch)
        torch.cuda.synchronize()
        a6, ast = a.out6.clone(), a.status.clone()
        b = match_batch(_plan(TRUSEQ, 0.1, 3, scan=False), batch)
        torch.cuda.synchronize()
        assert torch.equal(ast, b.status), gen
        assert torch.equal(a6, b.out6), gen
        assert int((ast == 1).sum()) > 100_000


def test_scan_ragged_and_views(hip, orc):
    """ragged batch + sub-sequence views (the second stage of linked adapters) through the scan"""
    import torch
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(5)
    reads = _reads(rng, TRUSEQ, 5000, 160, 0.05, 0.01)
    seqs, offsets = orc.pack_reads(reads)
    batch = ReadBatch.from_host(seqs, offsets)
    starts = torch.tensor([rng.randint(0, len(r)) for r in reads], dtype=torch.int64, device=batch.device)
    lens = torch.tensor([rng.randint(0, len(r) - int(s)) for r, s in zip(reads, starts.tolist())], dtype=torch.int32,
                        device=batch.device)
    view = batch.view(starts, lens)
    res = match_batch(_plan(TRUSEQ, 0.1, 3), view)
    got6, got_st, _ = res.cpu()
    sub = [r[int(s):int(s) + int(l)] for r, s, l in zip(reads, starts.tolist(), lens.tolist())]
    vs, vo = orc.pack_reads(sub)
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    oa = orc.Aligner(TRUSEQ, 0.1, 14, False, False, 1, 3)
    of = orc.KmerFinder(create_positions_and_kmers(TRUSEQ, 3, 0.1, back_adapter=True, front_adapter=False))
    w6, wst = orc.match_batch(oa, of, vs, vo)
    _same(got6, got_st, w6, wst, "views")


def test_one_plan_on_every_visible_device(hip, orc):
    """plans replicate their tables per device on first use (api.cpp plan_on_device): the same plan object
    serves every GPU of the node from one process -- each device matches its own shard of the reads"""
    import torch
    from cutadapt_amd import _lib
    from cutadapt_amd.batch import ReadBatch, match_batch
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    n_dev = _lib.device_count()
    assert n_dev >= 1
    plan = _plan(TRUSEQ, 0.1, 3)
    oa = orc.Aligner(TRUSEQ, 0.1, 14, False, False, 1, 3)
    of = orc.KmerFinder(create_positions_and_kmers(TRUSEQ, 3, 0.1, back_adapter=True, front_adapter=False))
    per = 30_000
    results = []
    for d in range(n_dev):
        with torch.cuda.device(d):
            batch = ReadBatch.synthetic(per, 150, [TRUSEQ], seed=9, first_index=d * per, device=torch.device("cuda", d))
            results.append((d, match_batch(plan, batch)))
    for d, res in results:
        with torch.cuda.device(d):
            torch.cuda.synchronize()
            got6, got_st, _ = res.cpu()
        seqs, offsets = orc.synth_reads(9, d * per, per, 150, [TRUSEQ])
        w6, wst = orc.match_batch(oa, of, seqs, offsets)
        _same(got6, got_st, w6, wst, f"device {d}")


def test_straggler_list_overflow_is_harmless(hip, orc):
    """a wave that finds the straggler list full must leave nothing behind that the second launch then reads.
    A tiny list (CAH_SCAN_RETRY_CAP), every wave shedding as early as it can (CAH_SCAN_RETRY=63), scratch full of
    garbage: the results must not change."""
    import os
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    ad = A.BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
    n = 300_000
    batch = ReadBatch.synthetic(n, 150, [TRUSEQ], seed=77, p_adapter=0.6)
    want = match_batch(ad._fused_plan, batch).cpu()
    for cap in ("1", "100", "5000"):
        os.environ["CAH_SCAN_RETRY"] = "63"
        os.environ["CAH_SCAN_RETRY_CAP"] = cap
        try:
            b2 = ReadBatch(batch.seqs, batch.offsets, validated=True)
            b2.workspace().fill_(0x7F)                       # queue slots never written read as 0x7F7F7F7F
            got = match_batch(ad._fused_plan, b2).cpu()
        finally:
            os.environ.pop("CAH_SCAN_RETRY", None)
            os.environ.pop("CAH_SCAN_RETRY_CAP", None)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), cap


def plan_sets(adapter, min_overlap, rate):
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    return create_positions_and_kmers(adapter.upper(), min_overlap, rate, back_adapter=True, front_adapter=False)
