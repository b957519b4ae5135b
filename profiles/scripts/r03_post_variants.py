#!/usr/bin/env python3
"""all-device way of the device FASTQ path: what the modifiers behind the adapter step cost, one at a time"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cutadapt_amd import workloads
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
n = 12_000_000
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
for name, opts in (("plain", {}), ("-q", dict(quality_cutoff=(0, 10))), ("--poly-a", dict(poly_a=True)), ("--max-ee", dict(max_expected_errors=5.0)),
                   ("-l", dict(length=100)), ("-q --poly-a --max-ee -m", dict(quality_cutoff=(0, 10), poly_a=True, max_expected_errors=5.0, minimum_length=20))):
    trim_fastq_gpu(fastq[:317 * 1000000], None, [ad], threads=4, **opts)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        st = trim_fastq_gpu(fastq, None, [ad], threads=4, **opts)
        best = min(best, time.perf_counter() - t0)
    print(name, round(n / best / 1e6, 1), "Mreads/s", st["way"], file=sys.stderr)
