#!/usr/bin/env python3
"""Where the general way of the device FASTQ path spends its host time (cProfile of one worker thread's chunks)."""
import cProfile, io, pstats, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cutadapt_amd import workloads
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
n = 2_000_000
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
trim_fastq_gpu(fastq[:317 * 200000], None, [ad], threads=1, **opts)
from cutadapt_amd import gpu_pipeline as gp
from cutadapt_amd.pipeline import BatchTrimmer
w = gp._take_worker(None, [], dev, {})
trimmer = BatchTrimmer([ad], device=dev, **opts)
chunks = list(gp._chunks(fastq, 64 << 20))
job = lambda chunk: trimmer.process_chunk(chunk, False, False, None, None, None)
w.run_general(chunks[0][0], job)
pr = cProfile.Profile()
pr.enable()
import time
t0 = time.perf_counter()
for data, fin in chunks:
    body, bufs = w.run_general(data, job)
    for b in bufs:
        w.pool.put(b)
dt = time.perf_counter() - t0
pr.disable()
print("worker in the main thread:", n / dt / 1e6, "Mreads/s")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:7000])
