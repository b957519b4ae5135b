#!/usr/bin/env python3
"""End to end for read pairs: two FASTQ byte streams in host memory -> two trimmed FASTQ streams (no sink), one MI355X."""
import io, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cutadapt_amd import workloads
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.gpu_pipeline import trim_fastq_gpu_paired
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
dev = torch.device("cuda", 0)
files = []
for mate in (0, 1):
    batch = workloads.device_batch("C5", n, mate=mate, device=dev)
    seqs = batch.seqs.view(n, 150).cpu().numpy()
    rec = np.empty((n, 317), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1:12] = ord("r"); rec[:, 12] = 10
    rec[:, 13:163] = seqs
    rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
    rec[:, 166:316] = ord("I"); rec[:, 316] = 10
    files.append(rec.reshape(-1).tobytes())
spec = workloads.SPECS["C5"]
class Null:
    def write(self, b): pass
out = []
for label, extra1, extra2, top in (
        ("all-device: -q 0,10, two adapters per mate, -m 20", dict(quality_cutoff=(0, 10)), dict(quality_cutoff=(0, 10)), dict(minimum_length=20)),
        ("general: the same with --times 2", dict(quality_cutoff=(0, 10), times=2), dict(quality_cutoff=(0, 10), times=2), dict(minimum_length=20))):
    def run():
        r1 = dict(adapters=[BackAdapter(s, max_errors=0.1, min_overlap=3) for s in spec["adapters"]], **extra1)
        r2 = dict(adapters=[BackAdapter(s, max_errors=0.1, min_overlap=3) for s in spec["adapters2"]], **extra2)
        return trim_fastq_gpu_paired(io.BytesIO(files[0]), io.BytesIO(files[1]), Null(), Null(), r1, r2, threads=4, **top)
    run()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); st = run(); best = min(best, time.perf_counter() - t0)
    out.append({"what": label, "Mpairs_per_s": n / best / 1e6, "way": st["way"], "pairs_written": st["pairs_written"]})
    print(out[-1], file=sys.stderr)
print(json.dumps(out, indent=1))
