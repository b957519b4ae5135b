#!/usr/bin/env python3
"""The general way of the device FASTQ path against the number of worker threads (host numpy between the kernels)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cutadapt_amd import workloads
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
n = 8_000_000
dev = torch.device("cuda", 0)
batch = workloads.device_batch("C2", n, device=dev)
seqs = batch.seqs.view(n, 150).cpu().numpy()
rec = np.empty((n, 317), dtype=np.uint8)
rec[:, 0] = ord("@"); rec[:, 1:12] = ord("r"); rec[:, 12] = 10
rec[:, 13:163] = seqs
rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
rec[:, 166:316] = ord("I"); rec[:, 316] = 10
fastq = torch.from_numpy(rec.reshape(-1)).pin_memory()
ad = BackAdapter(workloads.TRUSEQ_R1, max_errors=0.1, min_overlap=3)
opts = dict(quality_cutoff=(0, 10), poly_a=True)
out = []
for threads, chunk in ((4, 64), (8, 64), (12, 64), (16, 64), (8, 16), (16, 16)):
    trim_fastq_gpu(fastq[:317 * 400000], None, [ad], threads=threads, chunk_bytes=chunk << 20, **opts)
    t0 = time.perf_counter()
    st = trim_fastq_gpu(fastq, None, [ad], threads=threads, chunk_bytes=chunk << 20, **opts)
    dt = time.perf_counter() - t0
    out.append({"threads": threads, "chunk_MiB": chunk, "Mreads_per_s": n / dt / 1e6, "way": st["way"]})
    print(out[-1], file=sys.stderr)
print(json.dumps(out))
