#!/usr/bin/env python3
"""Where does the time of the host-fed read-PAIR path go?  Two FASTQ files (page cache) -> two trimmed streams (no sink),
one MI355X, all-device way.  (1) the reader alone (gpu_pipeline._paired_pieces: both files read and cut into pieces of
equal record counts), into pinned and into pageable buffers; (2) the whole pipeline with the stage seconds trim_fastq_gpu_paired
reports, reader buffers pinned (the product) against pageable (round 3's way: every piece is copied once more into a worker's
pinned buffer).  Usage: r04_paired_stages.py [n_pairs]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cutadapt_amd import workloads, gpu_pipeline as gp, pipeline as hp
from cutadapt_amd.adapters import BackAdapter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
dev = torch.device("cuda", 0)
tmp = "/dev/shm/cah_paired" if os.path.isdir("/dev/shm") else os.path.join(ROOT, "gpurun_out", "tmp")
os.makedirs(tmp, exist_ok=True)
paths = []
for mate in (0, 1):
    path = os.path.join(tmp, f"mate{mate + 1}.fastq")
    with open(path, "wb") as f:
        step = 4_000_000
        for lo in range(0, n, step):
            cnt = min(step, n - lo)
            batch = workloads.device_batch("C5", cnt, first_index=lo, mate=mate, device=dev)
            rec = torch.empty((cnt, 317), dtype=torch.uint8, device=dev)
            rec[:, 0] = ord("@"); rec[:, 1:12] = ord("r"); rec[:, 12] = 10
            rec[:, 13:163] = batch.seqs.view(cnt, 150)
            rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
            rec[:, 166:316] = ord("I"); rec[:, 316] = 10
            rec.cpu().numpy().tofile(f)
            del rec, batch
    paths.append(path)
file_gb = 2 * os.path.getsize(paths[0]) / 1e9
spec = workloads.SPECS["C5"]
out = {"pairs": n, "input_GB": file_gb, "host_cpus": os.cpu_count(), "files": tmp}


class Null:
    def write(self, b): pass


class Pageable:
    """round 3's buffers: pipeline.POOL (pageable); the worker stages every piece through its own pinned buffer"""
    get = staticmethod(hp.POOL.get)
    put = staticmethod(hp.POOL.put)
    @staticmethod
    def tensor_of(a): return a
    @staticmethod
    def trim(keep=8): pass


def reader_alone(pool, threads):
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        pieces = 0
        for d1, d2 in gp._paired_pieces(paths[0], paths[1], 32 << 20, threads=threads, pool=pool):
            pieces += 1
            pool.put(d1); pool.put(d2)
        best = min(best, time.perf_counter() - t0)
    return {"s": best, "GB_per_s": file_gb / best, "Mpairs_per_s": n / best / 1e6, "pieces": pieces}


def run(workers, chunk=None):
    r1 = dict(adapters=[BackAdapter(s, max_errors=0.1, min_overlap=3) for s in spec["adapters"]], quality_cutoff=(0, 10))
    r2 = dict(adapters=[BackAdapter(s, max_errors=0.1, min_overlap=3) for s in spec["adapters2"]], quality_cutoff=(0, 10))
    kw = {} if chunk is None else {"chunk_bytes": chunk}
    return gp.trim_fastq_gpu_paired(paths[0], paths[1], Null(), Null(), r1, r2, threads=workers, minimum_length=20, **kw)


def _timed(f):
    t0 = time.perf_counter(); f(); return time.perf_counter() - t0


pinned = gp._PINNED_INPUT
out["reader_alone"] = {}
for name, pool in (("pinned", pinned), ("pageable", Pageable)):
    for t in (4, 8):
        out["reader_alone"][f"{name}, {t} threads per file"] = reader_alone(pool, t)
        print(name, t, out["reader_alone"][f"{name}, {t} threads per file"], file=sys.stderr, flush=True)
out["pipeline"] = {}
for name, pool in (("pinned reader buffers", pinned), ("pageable reader buffers (round 3)", Pageable)):
    gp._PINNED_INPUT = pool
    for workers in (6, 12):
        run(workers)
        best, st_best = 1e9, None
        for _ in range(2):
            t0 = time.perf_counter(); st = run(workers); dt = time.perf_counter() - t0
            if dt < best: best, st_best = dt, st
        row = {"Mpairs_per_s": n / best / 1e6, "GB_per_s_in": file_gb / best, "way": st_best["way"], "pairs_written": st_best["pairs_written"],
               "stages": st_best.get("stages")}
        out["pipeline"][f"{name}, {workers} workers"] = row
        print(name, workers, row, file=sys.stderr, flush=True)
gp._PINNED_INPUT = pinned
# block size and reader threads (pinned buffers, 6 workers)
out["variants"] = {}
orig = gp._paired_pieces
for chunk in (64 << 20,):
    for rt in (3, 4, 6):
        gp._paired_pieces = lambda a, b, c, pool=None, _t=rt: orig(a, b, c, threads=_t, pool=pool)
        run(6, chunk)
        best = min(_timed(lambda: run(6, chunk)) for _ in range(2))
        out["variants"][f"{chunk >> 20} MiB blocks, {rt} reader threads per file"] = n / best / 1e6
        print(chunk >> 20, rt, n / best / 1e6, file=sys.stderr, flush=True)
gp._paired_pieces = orig
for p in paths:
    os.unlink(p)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r04p"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r04p", "paired_stages.json"), "w").write(json.dumps(out, indent=1))
