#!/usr/bin/env python3
"""Developer tool: where a wave of k_filter_stream2 spends its cycles.  Needs a library built with -DS2_TRACE
(python -c "from cutadapt_amd import build; build.build_library(extra_flags=['-DS2_TRACE'], out_path='cutadapt_amd/libcutadapt_hip_trace.so')")
and CAH_LIB_PATH pointing at it.  Stations: 0 piece start, 1 first half in the slot (loads waited for), 2 = 1 (no
prefetch in phase 0), 3 first half matched, 4 second half in the slot, 5 next piece's loads issued, 6 second half matched,
7 emitted."""
import ctypes as C
import sys
import numpy as np
import torch
from cutadapt_amd import _lib
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.batch import ReadBatch, match_batch

TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ad = BackAdapter(TRUSEQ, max_errors=0.1, min_overlap=3)
batch = ReadBatch.synthetic(n, 150, [TRUSEQ], seed=2)
if len(sys.argv) > 2 and sys.argv[2] == "views":            # the RV form: reads cut to 30 .. 150 characters, as views
    from cutadapt_amd import workloads
    batch = workloads.ragged_view_batch(batch)
for _ in range(2):
    match_batch(ad._fused_plan, batch)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_uint64 * (64 * 12))()
assert L.cah_debug_s2_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 12).astype(np.int64)
fine = t
t = t[:, :8]
d = np.diff(t, axis=1)
names = ["wait+slot0", "-", "match1", "slot1", "clear+issue", "match2", "emit"]
print("cycles per piece (s_memtime ticks), pieces 8..40 of one wave:")
sel = t[8:40]
for i, nm in enumerate(names):
    print(f"  {nm:12s} mean {d[8:40, i].mean():9.0f}  min {d[8:40, i].min():7d}  max {d[8:40, i].max():7d}")
print("  piece total  mean", np.diff(t[8:41, 0]).mean(), " (next piece start - this piece start)")
gap = t[1:41, 0] - t[0:40, 7]
print("  gap between a piece's last station and the next piece's first (every 8th: tile flush + barriers):")
print("   ", gap.tolist())
buf2 = (C.c_uint64 * (16 * 8))()
assert L.cah_debug_s2_trace_tile(buf2) == 0
tt = np.frombuffer(buf2, dtype=np.uint64).reshape(16, 8).astype(np.int64)
print("tile stations (cycles): init barriers | pieces | wait at the end barrier | flush")
for k in range(2, 10):
    print("   ", tt[k, 1] - tt[k, 0], tt[k, 2] - tt[k, 1], tt[k, 3] - tt[k, 2], tt[k, 4] - tt[k, 3])

f = fine[8:40]
print("finer: matched -> clear stores issued", (f[:, 8] - f[:, 5]).mean(), "| take_piece", (f[:, 9] - f[:, 8]).mean(), "| prefetch issue", (f[:, 6] - f[:, 9]).mean(),
      "| buffer wait", (f[:, 10] - f[:, 6]).mean(), "| emit", (f[:, 7] - f[:, 10]).mean())

st = lambda a, b: (f[:, b] - f[:, a]).mean()
print("stations: wait+slot0", st(0, 1), "| match1", st(2, 3), "| slot1", st(3, 4), "| match2", st(4, 5), "| clear", st(5, 8),
      "| take", st(8, 9), "| prefetch issue", st(9, 6), "| buffer wait", st(6, 10), "| emit", st(10, 7),
      "| piece", np.diff(fine[8:41, 0]).mean())
