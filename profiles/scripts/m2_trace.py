#!/usr/bin/env python3
"""Developer tool: where a wave of k_multi_stream spends its cycles (library built with -DM2_TRACE, CAH_LIB_PATH).
Stations: 0 piece start, 1 first half in the slot, 2 first half walked, 3 second half in the slot, 4 second half
walked, 5 class W resolved, 6 / 7 / 8 after the hi / lo / REF-only sweep (resolved), 9 next piece's loads issued;
10: resolve rounds | events << 32, 11: directory-walk iterations."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from cutadapt_amd import _lib, workloads
from cutadapt_amd import adapters as A
from cutadapt_amd.batch import ReadBatch, match_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ads = workloads.SPECS["C4"]["adapters"]
plan = _lib.Plan([A.BackAdapter(s, max_errors=0.1, min_overlap=3).matcher_spec() for s in ads])
batch = workloads.device_batch("C4", n)
match_batch(plan, batch)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_uint64 * (64 * 16))()
assert L.cah_debug_m2_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
sel = slice(4, 40)
d = np.diff(t[:, :10], axis=1)
names = ["wait+slot0", "walk half 1", "slot1", "walk half 2", "drain W", "sweep hi", "sweep lo", "sweep short", "issue next"]
for i, nm in enumerate(names):
    print(f"  {nm:12s} mean {d[sel, i].mean():9.0f}  min {d[sel, i].min():7d}  max {d[sel, i].max():7d}")
print("  piece total mean", np.diff(t[4:41, 0]).mean())
print("  resolve cycles per piece (x launches): setup", t[sel, 12].mean(), "through the directory walk", t[sel, 15].mean(), "through the emission", t[sel, 13].mean(), "through the flush", t[sel, 14].mean())
print("  rounds per piece", (t[sel, 10] & 0xFFFFFFFF).mean(), "events per piece", (t[sel, 10] >> 32).mean(), "walk iterations", t[sel, 11].mean())
