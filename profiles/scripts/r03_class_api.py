#!/usr/bin/env python3
"""C2 through the adapter class (BackAdapter.match_to_batch -> lazy BatchMatches) against batch.match_batch on preallocated
outputs: what the class API costs on top of the library call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cutadapt_amd import workloads
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.batch import BatchResult, match_batch
n = 100_000_000
dev = torch.device("cuda", 0)
batch = workloads.device_batch("C2", n, device=dev)
ad = BackAdapter(workloads.TRUSEQ_R1, max_errors=0.1, min_overlap=3)
out = BatchResult(torch.empty((n, 6), dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.uint8, device=dev),
                  torch.empty(n, dtype=torch.int32, device=dev))
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
a, _ = timed(lambda: match_batch(ad._fused_plan, batch, out))
b, bm = timed(lambda: ad.match_to_batch(batch))
c, _ = timed(lambda: int(ad.match_to_batch(batch).device_found().sum().item()))
print(f"batch.match_batch, preallocated outputs: {a:.2f} ms per 100 M reads = {n / a / 1e6:.2f} Greads/s")
print(f"BackAdapter.match_to_batch (device-resident BatchMatches): {b:.2f} ms = {n / b / 1e6:.2f} Greads/s ({(b / a - 1) * 100:+.1f} %)")
print(f"... + the number of matches counted on the device and brought back: {c:.2f} ms ({(c / a - 1) * 100:+.1f} %)")
