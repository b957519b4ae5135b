"""k_filter_stream (equally long short reads: the wave's 64 reads copied HBM -> LDS with full-width loads) against
the oracle and against the per-lane kernels (CAH_NO_STREAM=1) -- through the C ABI like every GPU test.

Covered: every length around the instances' limits (111/112, 159/160) and the chunk / tail-window boundaries,
batches that end inside a wave, inside a tile and after several tiles per block, a batch whose first read does not
start at byte 0 of an unaligned buffer, 3', 5' and anywhere adapters (lead, tail and head words), invalid bytes,
MODE 0 (kmers_present_batch) and MODE 1 (the survivor queue feeding scan + DP).
Reference semantics: src/cutadapt/_kmer_finder.pyx:170-257 (kmers_present), adapters.py:815-832 (match_to).
"""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"


def rs(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from cutadapt_amd import _lib
    _lib.lib()
    return _lib


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    o.build()
    return o


def make_reads(rng, n, count, adapter, p_adapter=0.5):
    reads = []
    m = len(adapter)
    for _ in range(count):
        r = list(rs(rng, n))
        if n and rng.random() < p_adapter:
            p = rng.randint(max(0, n - m - 5), n - 1) if rng.random() < 0.5 else rng.randint(0, n - 1)
            piece = list(adapter[:rng.randint(1, m)])
            for _e in range(rng.choice([0, 0, 0, 1, 2])):
                piece[rng.randrange(len(piece))] = rng.choice("ACGT")
            r[p:p + len(piece)] = piece
            r = r[:n]
        if n and rng.random() < 0.05:
            r[rng.randrange(n)] = rng.choice("Nn")
        reads.append("".join(r))
    return reads


class no_stream:
    def __enter__(self):
        os.environ["CAH_NO_STREAM"] = "1"

    def __exit__(self, *a):
        os.environ.pop("CAH_NO_STREAM", None)


def oracle_expect(orc, ad, reads):
    from cutadapt_amd.adapters import Where
    flags = int({"BackAdapter": Where.BACK, "FrontAdapter": Where.FRONT, "AnywhereAdapter": Where.ANYWHERE}[type(ad).__name__])
    finder = orc.KmerFinder(ad.kmer_finder.positions_and_kmers, ad.adapter_wildcards, ad.read_wildcards)
    oal = orc.Aligner(ad.sequence, ad.max_error_rate, flags=flags, wildcard_ref=ad.adapter_wildcards,
                      wildcard_query=ad.read_wildcards, indel_cost=1 if ad.indels else 100000,
                      min_overlap=ad.min_overlap)
    return [oal.locate(r) if finder.kmers_present(r) else None for r in reads], finder


def test_stream_lengths_vs_oracle_and_per_lane_kernels(hip, orc):
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(4242)
    ad = A.BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
    lengths = sorted({0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 34, 36, 50, 63, 64, 65, 75, 100, 101, 110, 111, 112, 113,
                      125, 127, 128, 129, 144, 145, 149, 150, 151, 152, 158, 159, 160, 161, 170,
                      # the LONG form of the streaming prefilter (segments of 160 characters, up to 640), and beyond it
                      162, 175, 176, 177, 192, 200, 239, 240, 241, 250, 251, 255, 256, 257, 300, 301, 319, 320, 321, 336,
                      400, 479, 480, 481, 500, 639, 640, 641, 700})
    for n in lengths:
        count = rng.choice([1, 63, 64, 65, 300, 1024, 1025, 2500])
        reads = make_reads(rng, n, count, TRUSEQ)
        batch = ReadBatch.from_strings(reads)
        got = match_batch(ad._fused_plan, batch).cpu()
        with no_stream():
            ref = match_batch(ad._fused_plan, ReadBatch.from_strings(reads)).cpu()
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0]), n
        if count <= 300:
            exp, _ = oracle_expect(orc, ad, reads)
            for i, e in enumerate(exp):
                g = tuple(int(v) for v in got[0][i]) if got[1][i] == 1 else None
                assert g == e, (n, reads[i], g, e)


def test_stream_many_tiles_per_block_and_present_mode(hip, orc):
    """More tiles than blocks (several pieces per wave, the prefetch crossing tile borders), MODE 0."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(99)
    ad = A.BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
    L = hip.lib()
    for n, count in ((150, 1_500_000), (101, 1_200_037), (36, 900_001)):
        batch = ReadBatch.synthetic(count, n, [TRUSEQ], seed=rng.randint(1, 1 << 30))
        got = match_batch(ad._fused_plan, batch)
        pres = torch.empty(count, dtype=torch.uint8, device=batch.device)
        hip.check(L.cah_kmers_present_batch(ad._fused_plan.handle, 0, batch.seqs.data_ptr(), batch.offsets.data_ptr(),
                                            None, count, pres.data_ptr(), None))
        with no_stream():
            ref = match_batch(ad._fused_plan, batch)
            pres_ref = torch.empty(count, dtype=torch.uint8, device=batch.device)
            hip.check(L.cah_kmers_present_batch(ad._fused_plan.handle, 0, batch.seqs.data_ptr(),
                                                batch.offsets.data_ptr(), None, count, pres_ref.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(got.out6, ref.out6) and torch.equal(got.status, ref.status), (n, count)
        assert torch.equal(pres, pres_ref), (n, count)
        assert int(pres.sum()) > count // 10
        # a sample against the oracle
        seqs = batch.seqs[:2000 * n].cpu().numpy().tobytes().decode()
        reads = [seqs[i * n:(i + 1) * n] for i in range(2000)]
        exp, finder = oracle_expect(orc, ad, reads)
        g6, gs, _ = got.cpu()
        p = pres[:2000].cpu().numpy()
        for i, e in enumerate(exp):
            g = tuple(int(v) for v in g6[i]) if gs[i] == 1 else None
            assert g == e, (n, reads[i], g, e)
            assert bool(p[i]) == bool(finder.kmers_present(reads[i]))


def test_stream_unaligned_start_and_other_adapter_kinds(hip, orc):
    """The batch starts in the middle of an unaligned buffer (offsets[0] != 0, odd base address); 5' and anywhere
    adapters (head words) and random 3' adapters with wildcards take the same kernel."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(7)
    for it in range(10):
        m = rng.choice([8, 13, 20, 33, 40, 57, 64])
        seq = rs(rng, m, "ACGT") if it % 3 else rs(rng, m, "ACGTN")
        cls = [A.BackAdapter, A.FrontAdapter, A.AnywhereAdapter][it % 3]
        ad = cls(seq, max_errors=rng.choice([0.0, 0.1, 0.2]), min_overlap=rng.randint(1, 6),
                 read_wildcards=rng.random() < 0.2, indels=rng.random() < 0.8)
        n = rng.choice([30, 48, 75, 100, 112, 150, 159])
        count = rng.choice([200, 300])
        reads = make_reads(rng, n, count, seq)
        junk = rng.randint(1, 37)
        from cutadapt_amd.batch import pack_strings
        seqs, offsets = pack_strings(reads)
        buf = np.concatenate([np.frombuffer(rs(rng, junk).encode(), dtype=np.uint8), seqs])
        dev = torch.device("cuda", torch.cuda.current_device())
        big = torch.from_numpy(buf.copy()).to(dev)
        pad = rng.randint(1, 7)
        shifted = torch.empty(big.numel() + pad, dtype=torch.uint8, device=dev)
        shifted[pad:] = big
        view = shifted[pad:]                       # base address = allocation + pad (not 16-byte aligned)
        assert view.data_ptr() % 16 != 0 or pad % 16 == 0
        offs = torch.from_numpy(offsets + junk).to(dev)
        batch = ReadBatch(view, offs, validated=True)
        got = match_batch(ad._fused_plan, batch).cpu()
        exp, _ = oracle_expect(orc, ad, reads)
        for i, e in enumerate(exp):
            g = tuple(int(v) for v in got[0][i]) if got[1][i] == 1 else None
            assert g == e, (type(ad).__name__, seq, n, reads[i], g, e)


def test_stream_flags_invalid_bytes(hip):
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(5)
    ad = A.BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
    n, count = 150, 700
    reads = make_reads(rng, n, count, TRUSEQ)
    from cutadapt_amd.batch import pack_strings
    seqs, offsets = pack_strings(reads)
    seqs = seqs.copy()
    bad = sorted(rng.sample(range(count), 9))
    for i, r in enumerate(bad):
        seqs[r * n + [0, n - 1, rng.randrange(n)][i % 3]] = 0xC3
    batch = ReadBatch.from_host(seqs, offsets, validated=True)
    got = match_batch(ad._fused_plan, batch).cpu()
    st = got[1]
    assert sorted(np.nonzero(st == 2)[0].tolist()) == bad


def _survivors(batch, n):
    """the survivor queue cah_match_batch left in the batch's workspace (layout: api.cpp, "workspace layout"):
    {read index: key}"""
    import torch
    ws = batch.workspace()
    torch.cuda.synchronize()
    count = int(ws[256:264].view(torch.int64).item())
    qbytes = (4 * n + 255) & ~255
    queue = ws[1024:1024 + 4 * count].view(torch.int32).cpu().numpy()
    keys = ws[1024 + qbytes:1024 + qbytes + count].cpu().numpy()
    assert len(set(queue.tolist())) == count
    return dict(zip(queue.tolist(), keys.tolist()))


def test_stream_survivor_queue_is_exactly_kmers_present(hip, orc):
    """The prefilter is exact, not merely sufficient: the survivor queue holds exactly the reads whose kmers_present is
    true (the oracle's), under the key the cost scan's column skipping relies on (first 4-character group in which a
    k-mer that lies in its window ends, min 255) -- the same queue the round-2 kernels and the per-lane kernels leave."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(31337)
    cases = [(TRUSEQ, 0.1, 3, n) for n in (150, 151, 149, 100, 76, 36, 160, 81, 80, 17, 161, 250, 300, 321, 480, 640)]
    for _ in range(12):
        m = rng.choice([12, 20, 25, 33, 34, 40])
        cases.append((rs(rng, m), rng.choice([0.0, 0.1, 0.2]), rng.randint(1, 6), rng.choice([150, 100, 50, 125, 250, 301])))
    for seq, rate, ov, n in cases:
        ad = A.BackAdapter(seq, max_errors=rate, min_overlap=ov)
        count = rng.choice([700, 1300, 9000])
        reads = make_reads(rng, n, count, seq, p_adapter=0.6)
        finder = orc.KmerFinder(ad.kmer_finder.positions_and_kmers, ad.adapter_wildcards, ad.read_wildcards)
        batch = ReadBatch.from_strings(reads)
        match_batch(ad._fused_plan, batch)
        got = _survivors(batch, count)
        want = {i for i, r in enumerate(reads) if finder.kmers_present(r)}
        assert set(got) == want, (seq, rate, ov, n, sorted(set(got) ^ want)[:5])
        for env in ("CAH_NO_STREAM2", "CAH_NO_STREAM"):
            os.environ[env] = "1"
            try:
                b2 = ReadBatch.from_strings(reads)
                match_batch(ad._fused_plan, b2)
                ref = _survivors(b2, count)
            finally:
                os.environ.pop(env, None)
            assert ref == got, (env, seq, rate, ov, n, [(i, got[i], ref.get(i)) for i in got if got[i] != ref.get(i)][:5])


def test_uniform_entry_point_equals_the_offsets_entry_point(hip, orc):
    """cah_match_batch_uniform (no offsets array) against cah_match_batch on the same reads, for every kind of plan the
    library has a path for: stream2 / lean / general prefilter, cost scan and cell DP, comparers, anchored adapters,
    adapters beyond 64 characters, sequential and fused multi-adapter plans, reads longer than the streaming kernels take."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd import _lib
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(20250925)
    long_ad = rs(rng, 80)
    many = [rs(rng, 33) for _ in range(12)]
    plans = [
        ("back", A.BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)._fused_plan, TRUSEQ),
        ("front", A.FrontAdapter(TRUSEQ[:20], max_errors=0.1, min_overlap=3)._fused_plan, TRUSEQ[:20]),
        ("anywhere", A.AnywhereAdapter(TRUSEQ[:25], max_errors=0.15, min_overlap=4)._fused_plan, TRUSEQ[:25]),
        ("prefix", A.PrefixAdapter("ACGTACGTAC", max_errors=0.1)._fused_plan, "ACGTACGTAC"),
        ("suffix-noindels", A.SuffixAdapter("TTGACCAGT", max_errors=0.12, indels=False)._fused_plan, "TTGACCAGT"),
        ("long", A.BackAdapter(long_ad, max_errors=0.1, min_overlap=5)._fused_plan, long_ad),
        ("twelve", _lib.Plan([A.BackAdapter(s, max_errors=0.1, min_overlap=3).matcher_spec() for s in many]), many[3]),
        ("two", _lib.Plan([A.BackAdapter(s, max_errors=0.1, min_overlap=3).matcher_spec() for s in many[:2]]), many[1]),
    ]
    for name, plan, seq in plans:
        for n, count in ((150, 3000), (100, 700), (36, 1500), (200, 900), (15, 400)):
            reads = make_reads(rng, n, count, seq, p_adapter=0.6)
            batch = ReadBatch.from_strings(reads)
            assert batch.uniform_len == n
            got = match_batch(plan, batch).cpu()
            os.environ["CAH_NO_UNIFORM"] = "1"
            try:
                ref = match_batch(plan, ReadBatch.from_strings(reads)).cpu()
            finally:
                os.environ.pop("CAH_NO_UNIFORM", None)
            assert np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0]), (name, n)
            if got[2] is not None and ref[2] is not None:
                assert np.array_equal(got[2][got[1] == 1], ref[2][ref[1] == 1]), (name, n)


def test_suffix_views_of_a_uniform_batch(hip, orc):
    """cah_match_batch_suffix_views (the second stage of a linked adapter on equally long reads: view r starts skip[r]
    characters into read r and ends where it ends): tuples against the oracle on the suffixes themselves and against the
    plain view entry point, and the survivor queue = exactly the suffixes whose kmers_present is true.  Skips of every
    size: 0, a multiple of 4, odd ones, the whole read."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(777)
    cases = [(TRUSEQ, 0.1, 3, 150), (TRUSEQ, 0.1, 3, 151), (TRUSEQ, 0.1, 3, 100), (TRUSEQ, 0.1, 3, 40), (TRUSEQ[:20], 0.2, 2, 76)]
    for _ in range(6):
        cases.append((rs(rng, rng.choice([12, 25, 33])), rng.choice([0.0, 0.1]), rng.randint(1, 5), rng.choice([150, 125, 50])))
    for seq, rate, ov, n in cases:
        ad = A.BackAdapter(seq, max_errors=rate, min_overlap=ov)
        count = rng.choice([500, 1500, 9000])
        reads = make_reads(rng, n, count, seq, p_adapter=0.7)
        # adapter copies in FRONT of the view start must not be seen: plant some at the read's head
        reads = [(seq + r)[:n] if rng.random() < 0.2 else r for r in reads]
        skip = np.array([rng.choice([0, 0, 16, 16, 4, 8, 1, 3, 17, 33, n - 2, n, rng.randint(0, n)]) for _ in reads], dtype=np.int64)
        subs = [r[int(k):] for r, k in zip(reads, skip)]
        want, finder = oracle_expect(orc, ad, subs)
        batch = ReadBatch.from_strings(reads)
        assert batch.uniform_len == n
        sk = torch.from_numpy(skip).cuda()
        results = {}
        for mode in ("suffix", "views"):
            view = batch.view(sk, batch.lengths() - sk)
            if mode == "suffix":
                view.suffix_of_uniform = n
            else:
                view.within_uniform = None                   # the plain view entry point ...
                view.max_len = 10 ** 6                       # ... and its per-lane kernels (no frame: cah_match_batch_frames is not asked)
            res = match_batch(ad._fused_plan, view)
            out6, st = res.out6.cpu().numpy(), res.status.cpu().numpy()
            results[mode] = (out6, st, _survivors(view, count))
            for i, w in enumerate(want):
                if w is None:
                    assert st[i] == 0, (mode, seq, n, i, skip[i])
                else:
                    assert st[i] == 1 and tuple(out6[i]) == tuple(w), (mode, seq, n, i, skip[i], tuple(out6[i]), w)
        present = {i for i, r in enumerate(subs) if finder.kmers_present(r)}
        assert set(results["suffix"][2]) == present == set(results["views"][2]), (seq, n)
        # the streamed form may only UNDER-state the first hit (the scan then starts earlier), never over-state it
        ks, kv = results["suffix"][2], results["views"][2]
        assert all(ks[i] <= kv[i] for i in ks), [(i, ks[i], kv[i], skip[i]) for i in ks if ks[i] > kv[i]][:5]


def test_inner_views_of_a_uniform_batch(hip, orc):
    """cah_match_batch_views (reads of a sequencer's batch cut by a modifier in front of the adapter search: view r is any
    part of read r): the prefilter streams the parent's reads end-aligned.  Tuples against the oracle on the views
    themselves and against the plain view entry point, the survivor queue = exactly the views whose kmers_present is
    true.  Views of every shape: whole reads, empty ones, cut at either end or both, at lengths that are and are not
    multiples of 16; batches of less than a piece, of the slow first / last pieces only, and of many tiles."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(2025 + int(os.environ.get("CAH_TEST_SEED_OFFSET", "0")))
    cases = [(TRUSEQ, 0.1, 3, 150), (TRUSEQ, 0.1, 3, 151), (TRUSEQ, 0.1, 3, 100), (TRUSEQ, 0.1, 3, 40), (TRUSEQ, 0.1, 3, 160),
             (TRUSEQ, 0.1, 3, 16), (TRUSEQ, 0.1, 3, 10), (TRUSEQ, 0.1, 3, 33), (TRUSEQ[:20], 0.2, 2, 76)]
    for _ in range(6):
        cases.append((rs(rng, rng.choice([12, 25, 33])), rng.choice([0.0, 0.1]), rng.randint(1, 5), rng.choice([150, 125, 50, 81])))
    for seq, rate, ov, n in cases:
        ad = A.BackAdapter(seq, max_errors=rate, min_overlap=ov)
        count = rng.choice([1, 40, 500, 1500, 9000, 20000])
        reads = make_reads(rng, n, count, seq, p_adapter=0.7)
        # adapter copies OUTSIDE the view must not be seen: plant some at the read's head and at its very end
        reads = [(seq + r)[:n] if rng.random() < 0.2 else ((r + seq)[-n:] if rng.random() < 0.2 else r) for r in reads]
        start = np.zeros(count, dtype=np.int64)
        length = np.zeros(count, dtype=np.int64)
        for i in range(count):
            shape = rng.randrange(8)
            if shape == 0:
                a, b = 0, n                                   # the whole read
            elif shape <= 3:
                a, b = 0, rng.choice([0, 1, 3, 15, 16, 17, 30, n - 1, n - 16, n - 33, rng.randint(0, n)])   # cut at the 3' end
            elif shape == 4:
                a = rng.randint(0, n); b = n                  # a suffix
            else:
                a = rng.randint(0, n); b = rng.randint(a, n)
            b = min(max(b, a), n)
            start[i], length[i] = a, b - a
        subs = [r[int(a):int(a + l)] for r, a, l in zip(reads, start, length)]
        want, finder = oracle_expect(orc, ad, subs)
        batch = ReadBatch.from_strings(reads)
        assert batch.uniform_len == n
        st_d, ln_d = torch.from_numpy(start).cuda(), torch.from_numpy(length).cuda()
        results = {}
        for mode in ("inner", "views"):
            view = batch.view(st_d, ln_d)
            if mode == "inner":
                assert view.within_uniform == n
            else:
                view.within_uniform = None                   # the plain view entry point ...
                view.max_len = 10 ** 6                       # ... and its per-lane kernels (no frame: cah_match_batch_frames is not asked)
            res = match_batch(ad._fused_plan, view)
            out6, st = res.out6.cpu().numpy(), res.status.cpu().numpy()
            results[mode] = (out6, st, _survivors(view, count))
            for i, w in enumerate(want):
                if w is None:
                    assert st[i] == 0, (mode, seq, n, i, start[i], length[i])
                else:
                    assert st[i] == 1 and tuple(out6[i]) == tuple(w), (mode, seq, n, i, start[i], length[i], tuple(out6[i]), w)
        present = {i for i, r in enumerate(subs) if finder.kmers_present(r)}
        assert set(results["inner"][2]) == present == set(results["views"][2]), (seq, n, count)
        # the streamed form may only UNDER-state the first hit (the scan then starts earlier), never over-state it
        ks, kv = results["inner"][2], results["views"][2]
        assert all(ks[i] <= kv[i] for i in ks), [(i, ks[i], kv[i], start[i], length[i]) for i in ks if ks[i] > kv[i]][:5]


def test_linked_adapter_on_uniform_reads_fused_and_staged(hip, orc):
    """cah_linked_match_batch_uniform: an anchored 5' adapter that tolerates no error is folded into the 3' adapter's
    streaming prefilter (one pass: 5' comparison, views, 3' prefilter); every other combination runs its stages one after
    the other.  Both against the two-stage rule over oracle results (reference adapters.py:1215-1227) and against the
    plain entry points (CAH_NO_UNIFORM=1)."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, linked_match_batch
    from test_gpu_configs import oracle_single
    rng = random.Random(4242)
    fronts = [("NNNNNNNNACGTACGT", 0.1, True), ("ACGTACGTAC", 0.0, True), ("NNNNACGT" * 3, 0.05, True),   # fusable
              ("ACGTTGCATTGACCAGT", 0.15, False), ("N" * 30 + "ACGTAC", 0.1, False)]                          # errors / 36 characters
    for fseq, frate, fusable in fronts:
        for n in (150, 100, 40, 151, 15):
            front = A.PrefixAdapter(fseq, max_errors=frate)
            back = A.BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
            count = rng.choice([300, 2000, 7000])
            reads = []
            for _ in range(count):
                head = "".join(rng.choice("ACGT") if ch == "N" else ch for ch in fseq)
                if rng.random() < 0.3:
                    q = rng.randrange(len(head))
                    head = head[:q] + rng.choice("ACGTN") + head[q + 1:]
                body = make_reads(rng, n, 1, TRUSEQ, p_adapter=0.6)[0]
                r = (head + body)[:n] if rng.random() < 0.75 else body
                reads.append(r)
            if n >= 20:
                bad = list(reads[5]); bad[3] = "é"; reads[5] = None       # an invalid byte inside the 5' columns
            seqs = np.frombuffer("".join(r if r is not None else "x" * n for r in reads).encode("latin-1"), dtype=np.uint8).copy()
            if n >= 20:
                seqs[5 * n: 6 * n] = np.frombuffer(("ACGé" + "A" * (n - 4)).encode("latin-1"), dtype=np.uint8)
                reads[5] = None
            offsets = np.arange(count + 1, dtype=np.int64) * n
            outs = {}
            for mode in ("uniform", "staged", "plain"):
                if mode == "plain":
                    os.environ["CAH_NO_UNIFORM"] = "1"
                if mode == "staged":
                    os.environ["CAH_NO_LINKED_FUSE"] = "1"      # the one call, its stages one after the other
                try:
                    batch = ReadBatch.from_host(seqs, offsets)
                    f, b, view = linked_match_batch(front._fused_plan, back._fused_plan, batch)
                    torch.cuda.synchronize()
                    outs[mode] = (f.out6.cpu().numpy(), f.status.cpu().numpy(), b.out6.cpu().numpy(), b.status.cpu().numpy(),
                                  view.offsets.cpu().numpy(), view.lens.cpu().numpy())
                finally:
                    os.environ.pop("CAH_NO_UNIFORM", None)
                    os.environ.pop("CAH_NO_LINKED_FUSE", None)
            for other in ("staged", "plain"):
                for a, bb in zip(outs["uniform"], outs[other]):
                    assert np.array_equal(a, bb), (fseq, n, other)
            f6, fst, b6, bst, starts, vlens = outs["uniform"]
            valid = [i for i, r in enumerate(reads) if r is not None]
            good = [reads[i] for i in valid]
            fc, ff = oracle_single(orc, front, good)
            assert np.array_equal(fst[valid] == 1, ff) and np.array_equal(f6[valid][ff], fc[ff]), (fseq, n)
            assert (f6[valid][~ff] == 0).all()
            stop = np.where(ff, fc[:, 3], 0)
            assert np.array_equal(starts[valid], offsets[valid] + stop) and np.array_equal(vlens[valid], n - stop)
            bc, bf = oracle_single(orc, back, [r[int(k):] for r, k in zip(good, stop)])
            assert np.array_equal(bst[valid] == 1, bf) and np.array_equal(b6[valid][bf], bc[bf]), (fseq, n)
            if n >= 20:
                assert fst[5] == 2 and starts[5] == 5 * n and vlens[5] == n
            assert ff.sum() > 0.3 * len(good) or n < len(fseq)


def test_packed_ragged_batches_stream_in_frames(hip, orc):
    """Round 6: a packed ragged batch (an offsets array; reads of 0 .. 150 characters) through the single adapter's streaming
    prefilter -- cah_match_batch_frames: every read end-aligned in a frame of the longest one's length, the copy gathered
    from the reads' ends (k_filter_stream2's RV form, suffix_views == 3) -- against the per-lane kernels in full and the
    oracle on a sample; then the same reads as views scattered over a larger buffer."""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd import batch as B
    from cutadapt_amd.batch import ReadBatch, match_batch
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    for adapter, rate in ((TRUSEQ, 0.1), ("GATCGGAAGAGCACACGTCT", 0.15)):
        ad = A.BackAdapter(adapter, max_errors=rate, min_overlap=3)
        plan = ad._fused_plan
        n = 300_000
        parent = ReadBatch.synthetic(n, 150, [adapter], seed=61, p_adapter=0.6, p_edit=0.03, p_n=0.003)
        idx = torch.arange(n, dtype=torch.int64, device=parent.device)
        lens = ((idx * 2654435761 + 311) >> 6) % 151
        lens[::1000] = 0
        lens[0] = 150
        lens[n - 1] = 150
        starts = ((idx * 40503) >> 3) % (151 - lens)
        off = torch.zeros(n + 1, dtype=torch.int64, device=parent.device)
        torch.cumsum(lens, 0, out=off[1:])
        cols = torch.arange(150, device=parent.device)
        keep = cols[None, :] < lens[:, None]
        gather = (parent.offsets[:n] + starts)[:, None] + cols[None, :]
        packed = parent.seqs[gather[keep]].contiguous()
        pb = ReadBatch(packed, off, validated=True)
        assert B._frame_len(plan, pb) == 150
        got = match_batch(plan, pb)
        torch.cuda.synchronize()
        os.environ["CAH_NO_FRAMES"] = "1"
        try:
            want = match_batch(plan, ReadBatch(packed, off, validated=True))
            torch.cuda.synchronize()
        finally:
            os.environ.pop("CAH_NO_FRAMES", None)
        assert torch.equal(got.status, want.status) and torch.equal(got.out6, want.out6), adapter
        assert int((want.status == 1).sum()) > 0.2 * n
        m = 40_000
        sq, offs = packed[: int(off[m].item())].cpu().numpy(), off[: m + 1].cpu().numpy()
        oa = orc.Aligner(adapter, rate, 14, False, False, 1, 3)
        of = orc.KmerFinder(create_positions_and_kmers(adapter, 3, rate, back_adapter=True, front_adapter=False))
        w6, wst = orc.match_batch(oa, of, sq, offs)
        assert np.array_equal(got.status[:m].cpu().numpy(), wst) and np.array_equal(got.out6[:m].cpu().numpy(), w6), adapter
        vb = ReadBatch(parent.seqs, parent.offsets[:n] + starts, lens.to(torch.int32), n_reads=n, validated=True)
        vb.max_len = 150
        got2 = match_batch(plan, vb)
        torch.cuda.synchronize()
        assert torch.equal(got2.status, want.status) and torch.equal(got2.out6, want.out6), adapter


def test_every_adapter_class_through_the_frames_entry(hip):
    """match_batch hands every ragged batch of 65 536 reads or more to cah_match_batch_frames: plans the streaming prefilters
    do not take (5' and anywhere adapters, anchored ones, wildcards, no indels, linked stages) must fall back inside the
    library -- the same rows as the plain entry point (CAH_NO_FRAMES=1), class by class"""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd import batch as B
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(515)
    ad = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    short = "ACGTTGCATGCA"
    reads = []
    for i in range(70_000):
        ln = rng.randint(0, 150)
        r = "".join(rng.choice("ACGT") for _ in range(ln))
        if i % 3 == 0 and ln > 40:
            at = rng.randint(0, ln - 20)
            r = (r[:at] + ad + r[at:])[:ln]
        if i % 5 == 0 and ln > 20:
            r = short + r[len(short):]
        if i % 7 == 0 and ln > 20:
            r = r[:-len(short)] + short
        reads.append(r)
    adapters = [A.BackAdapter(ad, max_errors=0.1, min_overlap=3), A.FrontAdapter(short, max_errors=0.1), A.AnywhereAdapter(short),
                A.PrefixAdapter(short), A.SuffixAdapter(short), A.NonInternalFrontAdapter(short), A.NonInternalBackAdapter(ad),
                A.RightmostFrontAdapter(short), A.RightmostBackAdapter(ad), A.BackAdapter("AGATCNGAAGNNCACACGTC", max_errors=0.1),
                A.BackAdapter(ad, max_errors=0.1, indels=False), A.BackAdapter(ad, max_errors=0.2, read_wildcards=True)]
    for a in adapters:
        batch = ReadBatch.from_strings(reads)
        assert batch.max_len == 150 and B._frame_len(a._fused_plan, batch) == 150
        got = a.match_to_batch(batch)
        os.environ["CAH_NO_FRAMES"] = "1"
        try:
            want = a.match_to_batch(ReadBatch.from_strings(reads))
        finally:
            os.environ.pop("CAH_NO_FRAMES", None)
        assert np.array_equal(got.found, want.found) and np.array_equal(got.coords[got.found], want.coords[want.found]), type(a).__name__
        assert got.found.sum() > 100, type(a).__name__
