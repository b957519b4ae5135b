#!/usr/bin/env python3
"""Developer tool: where a wave of k_back_scan3 spends its cycles (library built with -DSCAN_TRACE, CAH_LIB_PATH)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cutadapt_amd import _lib
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.batch import ReadBatch, match_batch
TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
ad = BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
batch = ReadBatch.synthetic(100_000_000, 150, [TRUSEQ], seed=2)
for _ in range(2):
    match_batch(ad._fused_plan, batch)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_uint64 * (256 * 8))()
assert L.cah_debug_scan3_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
t = t[t[:, 4] > 0][4:]
d = np.diff(t[:, :5], axis=1)
print("sub-batches traced:", len(t))
for i, nm in enumerate(["pre-pass", "scan", "classify", "stores+list"]):
    print(f"  {nm:12s} mean {d[:, i].mean():9.0f}  min {d[:, i].min():7d}  max {d[:, i].max():7d}")
print("  pre-pass chunks: mean", t[:, 5].mean(), " cycles per chunk:", (d[:, 0] / np.maximum(t[:, 5], 1)).mean())
print("  lanes F / E / T per sub-batch: mean", (t[:, 6] % 100).mean(), ((t[:, 6] // 100) % 100).mean(), (t[:, 6] // 10000).mean(), " retried:", t[:, 7].mean())
print("  sub-batch total mean", (t[:, 4] - t[:, 0]).mean(), " start-to-start", np.diff(t[:, 0]).mean())
for row in t[:24]:
    print("   ", (np.diff(row[:5])).tolist(), "pre chunks", row[5], "F/E/T", row[6] % 100, (row[6] // 100) % 100, row[6] // 10000, "retry", row[7])
