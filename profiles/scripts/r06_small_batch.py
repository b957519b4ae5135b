"""per-kernel times of the streaming multi-adapter path on batches of the size a length bucket has (C4 plan)"""
import ctypes as C, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from cutadapt_amd import _lib, workloads
from cutadapt_amd import adapters as A
from cutadapt_amd.batch import ReadBatch, match_batch
L = _lib.lib()
ads = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in workloads.SPECS["C4"]["adapters"]]
plan = _lib.Plan([a.matcher_spec() for a in ads])
for n, ln in ((826_000, 150), (826_000, 90), (826_000, 40), (100_000, 90), (5_000_000, 90)):
    b = ReadBatch.synthetic(n, ln, workloads.SPECS["C4"]["adapters"], seed=4)
    match_batch(plan, b); torch.cuda.synchronize()
    L.cah_profile_reset(); L.cah_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(5):
        match_batch(plan, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    L.cah_profile_enable(0)
    ms = (C.c_double * _lib.PROF_N)(); la = (C.c_int64 * _lib.PROF_N)(); un = (C.c_int64 * _lib.PROF_N)()
    L.cah_profile_read(ms, la, un)
    print(n, ln, plan.multi_kind(ln), "wall %.3f ms" % (dt * 1e3), {k: round(ms[i] / 5, 3) for k, i in (("filter", 0), ("dp", 1), ("scan", 3), ("merge", 4))}, flush=True)
