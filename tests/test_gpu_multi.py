"""The fused multi-adapter path (one prefilter pass for all adapters, cost scan and cell DP over (read,
adapter) pairs, atomic-max merge) against the one-adapter-at-a-time path of the same library and against
the oracle applying MultipleAdapters' rule (reference adapters.py:1265-1286).  GPU only."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rs(rng, n, al="ACGT"):
    return "".join(rng.choice(al) for _ in range(n))


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def plan_for(adapters, rate, min_overlap, fused=True):
    from cutadapt_amd import _lib
    from cutadapt_amd import adapters as A
    ads = [A.BackAdapter(s, max_errors=rate, min_overlap=min_overlap) for s in adapters]
    with env(CAH_NO_MULTI=None if fused else "1", CAH_MULTI_MIN="2"):
        return _lib.Plan([a.matcher_spec() for a in ads]), ads


def oracle_multiple(orc, ads, seqs, offsets):
    n = len(offsets) - 1
    want6 = np.zeros((n, 6), dtype=np.int32)
    want_st = np.zeros(n, dtype=np.uint8)
    want_best = np.full(n, -1, dtype=np.int32)
    for idx, ad in enumerate(ads):
        oa = orc.Aligner(ad.sequence, ad.max_error_rate, 14, False, False, 1, ad.min_overlap)
        of = orc.KmerFinder(ad.kmer_finder.positions_and_kmers)
        c6, st = orc.match_batch(oa, of, seqs, offsets)
        f = st == 1
        better = f & ((want_st == 0) | (c6[:, 4] > want6[:, 4]) | ((c6[:, 4] == want6[:, 4]) & (c6[:, 5] < want6[:, 5])))
        want6[better] = c6[better]
        want_best[better] = idx
        want_st[better] = 1
    return want6, want_st, want_best


def test_fused_equals_sequential_at_scale(hip):
    import torch
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(96)
    for n_adapters, m, n_reads in ((96, 33, 1_500_000), (12, 20, 1_000_000), (2, 33, 2_000_000), (128, 12, 300_000)):
        seqs = [rs(rng, m) for _ in range(n_adapters)]
        batch = ReadBatch.synthetic(n_reads, 150, seqs, seed=40 + n_adapters, p_adapter=0.4, p_edit=0.04, p_n=0.01)
        fused, _ = plan_for(seqs, 0.1, 3, fused=True)
        plain, _ = plan_for(seqs, 0.1, 3, fused=False)
        # small pair capacity: the batch is processed in several chunks
        with env(CAH_MULTI_PAIR_CAP=n_adapters * 200_000):
            a = match_batch(fused, batch)
            torch.cuda.synchronize()
        a6, ast, ab = a.out6.clone(), a.status.clone(), a.best_adapter.clone()
        b = match_batch(plain, batch)
        torch.cuda.synchronize()
        assert torch.equal(ast, b.status), (n_adapters, m)
        assert torch.equal(a6, b.out6), (n_adapters, m)
        found = ast == 1
        assert torch.equal(ab[found], b.best_adapter[found]), (n_adapters, m)
        assert int(found.sum()) > 0.3 * n_reads


def test_fused_fuzz_vs_oracle(hip, orc):
    from cutadapt_amd.batch import ReadBatch, match_batch
    rng = random.Random(4096)
    total = 0
    for it in range(40):
        m = rng.randint(5, 48)
        n_adapters = rng.choice([2, 3, 7, 24, 60])
        rate = rng.choice([0.0, 0.05, 0.1, 0.2])
        min_overlap = rng.choice([1, 3, 5])
        seqs = [rs(rng, m) for _ in range(n_adapters)]
        if it % 4 == 0:                         # near-duplicate adapters: ties between adapters
            seqs[1] = seqs[0][:-1] + rng.choice("ACGT")
            seqs[-1] = seqs[0]
        reads = []
        for _ in range(1500):
            n = rng.randint(0, 170)
            s = [rng.choice("ACGT") for _ in range(n)]
            for _c in range(rng.randint(0, 2)):
                ad = list(rng.choice(seqs))
                for _e in range(rng.choice([0, 0, 1, 2, 4])):
                    if ad:
                        x = rng.randrange(len(ad))
                        op = rng.randint(0, 2)
                        if op == 0:
                            ad[x] = rng.choice("ACGT")
                        elif op == 1:
                            ad.insert(x, rng.choice("ACGT"))
                        else:
                            del ad[x]
                pos = rng.randint(0, n) if rng.random() < 0.6 else max(0, n - rng.randint(1, len(ad) + 1))
                s[pos:pos + len(ad)] = ad
                s = s[:n]
            s = "".join(s)
            if rng.random() < 0.2:
                s = "".join(c if rng.random() > 0.03 else rng.choice("Nnacgt.") for c in s)
            reads.append(s)
        sq, offs = orc.pack_reads(reads)
        plan, ads = plan_for(seqs, rate, min_overlap)
        batch = ReadBatch.from_host(sq, offs)
        with env(CAH_MULTI_PAIR_CAP=n_adapters * 500):
            res = match_batch(plan, batch)
            got6, got_st, got_best = res.cpu()
        want6, want_st, want_best = oracle_multiple(orc, ads, sq, offs)
        what = f"it {it} m {m} A {n_adapters} rate {rate} O {min_overlap}"
        assert np.array_equal(got_st, want_st), (what, np.nonzero(got_st != want_st)[0][:5])
        bad = np.nonzero((got6 != want6).any(axis=1))[0]
        assert len(bad) == 0, (what, bad[:5], got6[bad[:2]], want6[bad[:2]], reads[int(bad[0])])
        f = want_st == 1
        assert np.array_equal(got_best[f], want_best[f]), what
        total += len(reads)
    assert total >= 60_000


def test_fused_path_is_taken(hip):
    """the plan of 96 equal-length 3' adapters asks for the larger workspace (i.e. the fused path exists); mixed
    lengths or wildcard adapters do not"""
    from cutadapt_amd import _lib
    rng = random.Random(1)
    seqs = [rs(rng, 33) for _ in range(96)]
    plan, _ = plan_for(seqs, 0.1, 3)
    L = _lib.lib()
    assert L.cah_plan_workspace_bytes(plan.handle, 1000) > L.cah_workspace_bytes(1000)
    mixed, _ = plan_for(seqs[:5] + [rs(rng, 20)], 0.1, 3)
    assert L.cah_plan_workspace_bytes(mixed.handle, 1000) == L.cah_workspace_bytes(1000)
    off, _ = plan_for(seqs, 0.1, 3, fused=False)
    assert L.cah_plan_workspace_bytes(off.handle, 1000) == L.cah_workspace_bytes(1000)
