#!/usr/bin/env python3
"""Read pairs end to end from two FASTQ FILES (page cache) -> two trimmed streams (no sink), one MI355X, all-device way:
the reader cuts both files into pieces of equal record counts (gpu_pipeline._paired_pieces: preadv + line-feed counting
on sub-ranges by several threads per file).  Usage: r03_e2e_paired_files.py [n_pairs] [reader threads ...]"""
import io, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cutadapt_amd import workloads, gpu_pipeline as gp
from cutadapt_amd.adapters import BackAdapter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
reader_threads = [int(x) for x in sys.argv[2:]] or [1, 4, 8]
dev = torch.device("cuda", 0)
tmp = os.path.join(ROOT, "gpurun_out", "tmp")
os.makedirs(tmp, exist_ok=True)
paths = []
for mate in (0, 1):
    batch = workloads.device_batch("C5", n, mate=mate, device=dev)
    rec = torch.empty((n, 317), dtype=torch.uint8, device=dev)
    rec[:, 0] = ord("@"); rec[:, 1:12] = ord("r"); rec[:, 12] = 10
    rec[:, 13:163] = batch.seqs.view(n, 150)
    rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
    rec[:, 166:316] = ord("I"); rec[:, 316] = 10
    paths.append(os.path.join(tmp, f"mate{mate + 1}.fastq"))
    rec.cpu().numpy().tofile(paths[-1])
    del rec, batch
spec = workloads.SPECS["C5"]
class Null:
    def write(self, b): pass
def run(src):
    r1 = dict(adapters=[BackAdapter(s, max_errors=0.1, min_overlap=3) for s in spec["adapters"]], quality_cutoff=(0, 10))
    r2 = dict(adapters=[BackAdapter(s, max_errors=0.1, min_overlap=3) for s in spec["adapters2"]], quality_cutoff=(0, 10))
    return gp.trim_fastq_gpu_paired(src[0], src[1], Null(), Null(), r1, r2, threads=int(os.environ.get('WORKERS', '6')), minimum_length=20)
out = []
orig = gp._paired_pieces
for t in reader_threads:
    gp._paired_pieces = lambda a, b, c, _t=t: orig(a, b, c, threads=_t)
    run(paths)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); st = run(paths); best = min(best, time.perf_counter() - t0)
    out.append({"what": f"all-device, two files, {t} reader thread(s) per file, {os.environ.get('WORKERS', '6')} workers", "Mpairs_per_s": n / best / 1e6, "way": st["way"],
                "pairs_written": st["pairs_written"]})
    print(out[-1], file=sys.stderr, flush=True)
    open(os.path.join(ROOT, "gpurun_out", "r3", "e2e_paired_files" + os.environ.get("TAG", "") + ".json"), "w").write(json.dumps(out, indent=1))
for p in paths:
    os.unlink(p)
