#!/usr/bin/env python3
"""Open-ended fuzz of the streaming multi-adapter path's RULES (test infrastructure; CPU only): random plans through the host
model (tests/host_model/multi2_model.cpp = the product's multi2.h + back_scan.h under g++) against the oracle applying
MultipleAdapters' rule.  tests/test_multi2_model.py runs a fixed sample of this generator; this script runs as many plans as
asked, from any seed, in two parameter sets.  Round 4 ran 5 500 plans (seeds 21-32, 41-54, 61-72, 101-1212).
Usage: python tests/host_model/multi2_fuzz.py SEED PLANS [--wide]      (M2M_EXTEND=k: scan k chunks past a pair's window)"""
import os, random, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as orc
import test_multi2_model as T

seed, plans = int(sys.argv[1]), int(sys.argv[2])
wide = "--wide" in sys.argv
rng = np.random.default_rng(seed)
prng = random.Random(seed + 1)
model = T.model.__wrapped__()
built = precise = 0
for it in range(plans):
    m = int(rng.choice([12, 16, 20, 24, 25, 28, 30, 32, 33, 34, 35, 36, 40, 50, 64]))
    count = int(rng.choice([2, 3, 8, 24, 48]))
    ads = ["".join(prng.choice("ACGT") for _ in range(m)) for _ in range(count)]
    if it % 4 == 0 and count > 2:                                # near-duplicates, an exact duplicate
        ads[1] = ads[0][:-1] + prng.choice("ACGT"); ads[-1] = ads[0]
    if it % 7 == 3:                                              # repetitive adapters: chunks that repeat inside the adapter
        unit = "".join(prng.choice("ACGT") for _ in range(int(rng.choice([2, 3, 4, 5]))))
        ads[0] = (unit * m)[:m]
    if wide:
        rate = float(rng.choice([0.05, 0.1, 0.13, 0.17, 0.2, 0.3])); O = int(rng.choice([1, 2, 3, 4, 6, 10, 12]))
        n = int(rng.choice([16, 17, 18, 20, 24, 31, 32, 33, 47, 48, 49, 64, 65, 100, 143, 144, 145, 159, 160]))
        p_n = float(rng.choice([0.0, 0.03, 0.08]))
    else:
        rate = float(rng.choice([0.08, 0.1, 0.1, 0.12, 0.15, 0.2, 0.25])); O = int(rng.choice([1, 3, 5, 8]))
        n = int(rng.integers(16, 161))
        p_n = float(rng.choice([0.0, 0.01]))
    # reads that END with edited adapter prefixes (errors in the margins of the windows, second partial copies) ...
    reads = T.tail_reads(rng, ads, 1500, n, p_n=p_n)
    reads = [r if len(r) == n else (r + "A" * n)[:n] for r in reads]
    if wide:
        reads = [r.lower() if i % 5 == 0 else r for i, r in enumerate(reads)]
    # ... and reads with the adapter INSIDE (whole-read pairs: one and several chunk hits, many edits)
    sq2, of2 = orc.synth_reads(int(rng.integers(1, 10 ** 6)), 0, 1500, n, ads, p_adapter=float(rng.choice([0.3, 0.8])),
                               p_edit=float(rng.choice([0.03, 0.08, 0.12])), p_n=0.005)
    sq, offs = orc.pack_reads(reads)
    sq = np.concatenate([sq, sq2]); offs = np.concatenate([offs, of2[1:] + offs[-1]])
    st = T.run(model, ads, rate, O, sq, offs, f"seed {seed} plan {it} m {m} x {count} rate {rate} O {O} n {n}", must_build=False)
    if st is not None:
        built += 1; precise += int(st[7])
print(f"seed {seed}: {plans} plans ({built} take the streaming form, {precise} pairs on the window of one occurrence): all identical to the oracle")
