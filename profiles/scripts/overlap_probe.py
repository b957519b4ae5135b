#!/usr/bin/env python3
"""Does running two halves of a batch on two HIP streams (prefilter of one half beside cost scan + DP of the other)
beat one stream?  100 M x 150 bp C2 reads, cah_match_batch per half on its own torch stream.
Usage: python profiles/scripts/overlap_probe.py [n_reads] [parts,...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from cutadapt_amd import workloads
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.batch import BatchResult, ReadBatch, match_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
parts_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
dev = torch.device("cuda", 0)
batch = workloads.device_batch("C2", n, device=dev)
plan = BackAdapter(workloads.TRUSEQ_R1, max_errors=0.1, min_overlap=3)._fused_plan
L = 150
out = {"reads": n, "runs": []}
ref = None
for parts in parts_list:
    per = n // parts
    subs, outs = [], []
    for i in range(parts):
        lo, hi = i * per, (n if i == parts - 1 else (i + 1) * per)
        seqs = batch.seqs[lo * L: hi * L]
        offs = (batch.offsets[lo: hi + 1] - lo * L).contiguous()
        b = ReadBatch(seqs, offs, validated=True)
        b.workspace(plan)
        subs.append(b)
        k = hi - lo
        outs.append(BatchResult(torch.empty((k, 6), dtype=torch.int32, device=dev), torch.empty(k, dtype=torch.uint8, device=dev),
                                torch.empty(k, dtype=torch.int32, device=dev)))
    streams = [torch.cuda.Stream(device=dev) for _ in range(min(parts, 2))]

    def step():
        for i, (b, o) in enumerate(zip(subs, outs)):
            with torch.cuda.stream(streams[i % len(streams)]):
                match_batch(plan, b, o)

    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    st = torch.cat([o.status for o in outs])
    o6 = torch.cat([o.out6 for o in outs])
    if ref is None:
        ref = (st.clone(), o6.clone())
    same = bool(torch.equal(st, ref[0]) and torch.equal(o6, ref[1]))
    out["runs"].append({"parts": parts, "streams": len(streams), "ms_per_step": ms, "Greads_per_s": n / ms / 1e6, "same_results": same})
    print(out["runs"][-1], file=sys.stderr)
    del subs, outs
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
