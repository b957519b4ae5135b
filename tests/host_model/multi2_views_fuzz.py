#!/usr/bin/env python3
"""Open-ended fuzz of the PADDED-FRAME rules of the streaming multi-adapter path (views streamed end-aligned: multi2.hip RV form,
DESIGN 3.5) through the host model against the oracle on the views (test infrastructure; CPU only).  Round 6 ran seeds 1-3 x 80 plans.
Usage: python tests/host_model/multi2_views_fuzz.py SEED PLANS"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_multi2_model as T
from oracle import oracle as orc
model=T.model.__wrapped__()
seed=int(sys.argv[1]); its=int(sys.argv[2])
rng=np.random.default_rng(seed)
built=0
for it in range(its):
    m = int(rng.choice([16,20, 24, 28, 30, 33, 34, 36, 40, 50, 64]))
    count = int(rng.choice([2, 3, 8, 24, 48]))
    ads = T.rand_adapters(rng, count, m)
    if it % 5 == 0 and count > 2: ads[1] = ads[0][:-1] + ("A" if ads[0][-1] != "A" else "C")
    rate = float(rng.choice([0.08,0.1, 0.1, 0.12, 0.15, 0.2, 0.25])); O = int(rng.choice([1, 3, 5, 8]))
    N = int(rng.integers(40, 161))
    reads = T.tail_reads(rng, ads, 500, N, p_n=0.005)
    reads = [r if len(r) == N else (r + "A" * N)[:N] for r in reads]
    sq2, of2 = orc.synth_reads(int(rng.integers(1, 10 ** 6)), 0, 700, N, ads, p_adapter=0.85, p_edit=float(rng.choice([0.02, 0.06, 0.1])), p_n=0.005)
    reads += [bytes(sq2[of2[i]:of2[i + 1]]).decode("latin-1") for i in range(700)]
    views=[]
    for i, r in enumerate(reads):
        mode = i % 4
        a = int(rng.integers(0, N + 1)) if mode in (1, 3) else 0
        b = int(rng.integers(a, N + 1)) if mode in (2, 3) else N
        if mode == 1 and rng.random() < 0.6:
            for ad in ads:
                at = r.find(ad[4:12])
                if at >= 0:
                    a = min(N, max(0, at - 4 + int(rng.integers(0, 8)))); break
        views.append(r[a:b])
    seqs, offsets = orc.pack_reads(views)
    pads = (N - np.diff(offsets)).astype(np.int32)
    if T.run(model, ads, rate, O, seqs, offsets, f"views seed {seed} it {it} m {m} x {count} rate {rate} O {O} N {N}", must_build=False, pads=pads) is not None: built+=1
print(f"seed {seed}: {its} plans, {built} built: all identical to the oracle")
