#!/usr/bin/env python3
"""End to end, host-fed, second round-2 session: the two ways of bringing the trimmed FASTQ back -- formatted on the
GPU (assemble="device": ~280 bytes per record over PCIe outbound) or put together from the input chunk on the
worker's host core (assemble="host": 57 bytes per record outbound, one memcpy pass per chunk on the host).
Usage: python profiles/scripts/e2e_gpu2.py [n_reads] > profiles/r02/s2e_e2e_gpu.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from cutadapt_amd import workloads
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.gpu_pipeline import trim_fastq_gpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
CONFIGS = ((1, 64), (3, 64), (4, 64), (6, 64), (8, 64), (12, 64), (8, 32), (12, 32))
if len(sys.argv) > 2:
    CONFIGS = tuple(tuple(int(x) for x in c.split("x")) for c in sys.argv[2].split(","))
dev = torch.device("cuda", 0)
batch = workloads.device_batch("C2", n, device=dev)
seqs = batch.seqs.view(n, 150)
idx = torch.arange(n, dtype=torch.int64, device=dev)
rec = torch.empty((n, 317), dtype=torch.uint8, device=dev)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 12] = 10
for k in range(10):
    rec[:, 11 - k] = ((idx // (10 ** k)) % 10 + 48).to(torch.uint8)
rec[:, 13:163] = seqs
rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
rec[:, 166:316] = ord("I")
rec[:, 316] = 10
fastq = torch.empty(n * 317, dtype=torch.uint8).pin_memory()
fastq.copy_(rec.view(-1))
del rec, idx, batch, seqs
torch.cuda.empty_cache()
adapter = BackAdapter(workloads.TRUSEQ_R1, max_errors=0.1, min_overlap=3)
out = {"reads": n, "fastq_bytes": int(fastq.numel()), "record_bytes": 317, "host_cpus": os.cpu_count(), "runs": []}


def timed(label, fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["runs"].append({"what": label, "seconds": best, "Mreads_per_s": n / best / 1e6,
                        "GB_per_s_in": fastq.numel() / best / 1e9, "with_adapters": stats["with_adapters"],
                        "bytes_out": stats.get("bytes_out")})
    print(out["runs"][-1], file=sys.stderr)
    return stats


trim_fastq_gpu(fastq[: 317 * 200_000], None, [adapter], threads=2)          # warm-up
trim_fastq_gpu(fastq[: 317 * 200_000], None, [adapter], threads=2, assemble="host")
trim_fastq_gpu(fastq, None, [adapter], threads=14, assemble="mixed3")        # every worker's buffers exist
ref = None
MODES = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("device", "host")
for assemble in MODES:
    for threads, chunk_mib in CONFIGS:
        st = timed(f"assemble={assemble}: pinned input, no sink, {threads} worker thread(s) x {chunk_mib} MiB chunks",
                   lambda: trim_fastq_gpu(fastq, None, [adapter], threads=threads, chunk_bytes=chunk_mib << 20,
                                          assemble=assemble))
        if ref is None:
            ref = (st["reads"], st["with_adapters"], st["bp_out"], st["bytes_out"])
        assert (st["reads"], st["with_adapters"], st["bp_out"], st["bytes_out"]) == ref
devnull = open(os.devnull, "wb")
for assemble in MODES:
    timed(f"assemble={assemble}: pinned input, output written to /dev/null, 8 worker threads",
          lambda: trim_fastq_gpu(fastq, devnull, [adapter], threads=8, assemble=assemble))
print(json.dumps(out, indent=1))
