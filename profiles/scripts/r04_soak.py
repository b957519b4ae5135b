#!/usr/bin/env python3
"""Soak of the streaming multi-adapter path (k_multi_stream + k_multi_scan), beyond what the -m gpu suite runs:
(1) random plans (adapter length, count, rate, min_overlap, read length, pool size) x 3 000 reads against the ORACLE
    (tests/test_gpu_multi2.py's generator with other seeds);
(2) random plans x 4 M synthetic reads: the streaming form against the older kernels of the same library
    (k_multi_filter + k_back_scan<true>: an independent implementation), read for read.
Prints one line per case; exits non-zero on the first difference.  Usage: r04_soak.py [seconds for part 1] [cases of part 2]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle as orc
from test_gpu_multi import env, plan_for, rs
from test_gpu_multi2 import run_uniform
from test_multi2_model import tail_reads
from cutadapt_amd.batch import ReadBatch, match_batch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
cases2 = int(sys.argv[2]) if len(sys.argv) > 2 else 12
SEED = int(os.environ.get("SOAK_SEED", "40404"))          # (round 5: other seeds per run)
rng = np.random.default_rng(SEED)
prng = random.Random(SEED + 1)
t0 = time.time(); it = streamed = reads_total = 0
while time.time() - t0 < budget:
    m = int(rng.choice([12, 16, 20, 25, 30, 32, 33, 34, 35, 40, 50, 64]))
    count = int(rng.choice([2, 3, 8, 24, 48, 96, 128]))
    seqs = [rs(prng, m) for _ in range(count)]
    if it % 4 == 0 and count > 2:
        seqs[1] = seqs[0][:-1] + prng.choice("ACGT"); seqs[-1] = seqs[0]
    rate = float(rng.choice([0.1, 0.1, 0.12, 0.15, 0.2])); O = int(rng.choice([1, 3, 5, 8]))
    n = int(rng.integers(16, 161))
    reads = tail_reads(rng, seqs, 3000, n, p_n=float(rng.choice([0.0, 0.01])))
    reads = [r if len(r) == n else (r + "A" * n)[:n] for r in reads]
    plan, _ = plan_for(seqs, rate, O)
    kind = plan.multi_kind(n)
    streamed += kind == "stream"
    cap = count * int(rng.choice([500, 700, 3000])) if it % 3 == 0 else None
    found = run_uniform(orc, seqs, rate, O, reads, f"soak {it}", expect=None, pair_cap=cap)
    print(f"oracle case {it}: m {m} x {count} rate {rate} O {O} n {n} pool {cap} ({kind}): {found} of 3000 matched, identical", flush=True)
    it += 1; reads_total += 3000
print(f"part 1: {it} plans ({streamed} on the streaming form), {reads_total} reads: all identical to the oracle", flush=True)
for c in range(cases2):
    m = int(rng.choice([20, 25, 30, 33, 34, 40, 50, 64])); count = int(rng.choice([8, 24, 96, 128]))
    n = int(rng.choice([50, 75, 100, 125, 150, 151, 160])); n_reads = 4_000_000
    seqs = [rs(prng, m) for _ in range(count)]
    batch = ReadBatch.synthetic(n_reads, n, seqs, seed=900 + c, p_adapter=float(rng.choice([0.1, 0.4, 0.9])), p_edit=float(rng.choice([0.02, 0.06])), p_n=0.01)
    plan, _ = plan_for(seqs, 0.1, 3)
    kind = plan.multi_kind(n)
    cap = str(count * int(rng.choice([20_000, 200_000]))) if c % 2 else None
    with env(CAH_MULTI_PAIR_CAP=cap):
        a = match_batch(plan, batch); torch.cuda.synchronize()
    a6, ast, ab = a.out6.clone(), a.status.clone(), a.best_adapter.clone()
    with env(CAH_NO_MULTI2="1"):
        b = match_batch(plan, batch); torch.cuda.synchronize()
    ok = torch.equal(ast, b.status) and torch.equal(a6, b.out6) and torch.equal(ab[ast == 1], b.best_adapter[ast == 1])
    print(f"scale case {c}: m {m} x {count} n {n} pool {cap} ({kind}): {int((ast == 1).sum())} of {n_reads} matched, {'identical' if ok else 'DIFFERENT'}", flush=True)
    if not ok:
        sys.exit(1)
print("part 2: all identical to the older kernels")
