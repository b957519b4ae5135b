#!/usr/bin/env python3
"""Feeder scaling on ONE GPU (round 4): how much FASTQ can k feeders move when the device is not the
bound?  k = 1, 2, 4, 8 feeders against devices=[0]*k, as threads of one process and as one process each, with the device
work stubbed (``_stub_device``: the range read into pinned memory, the hand-over and the ordered sink are all there,
no kernel and no copy runs) -- and, for comparison, the real pipeline with 1 and 2 feeders.  The file sits in /dev/shm
(page cache: what a warm file system gives).  Usage: python profiles/scripts/r04_feeder_scaling.py [n_reads] > profiles/r04/feeder_scaling.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from cutadapt_amd import workloads
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.gpu_pipeline import trim_fastq_gpu


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
    REPEAT = int(sys.argv[2]) if len(sys.argv) > 2 else 12       # the file's pieces over and over: runs of seconds, not tenths
    dev = torch.device("cuda", 0)
    batch = workloads.device_batch("C2", n, device=dev)
    seqs = batch.seqs.view(n, 150)
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    rec = torch.empty((n, 317), dtype=torch.uint8, device=dev)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 12] = 10
    for k in range(10):
        rec[:, 11 - k] = ((idx // (10 ** k)) % 10 + 48).to(torch.uint8)
    rec[:, 13:163] = seqs
    rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
    rec[:, 166:316] = ord("I")
    rec[:, 316] = 10
    path = "/dev/shm/r04_feeder_scaling.fastq" if os.path.isdir("/dev/shm") else "/tmp/r04_feeder_scaling.fastq"
    with open(path, "wb") as f:
        f.write(memoryview(rec.view(-1).cpu().numpy()))
    size = os.path.getsize(path)
    del rec, idx, batch, seqs
    torch.cuda.empty_cache()
    adapter = BackAdapter(workloads.TRUSEQ_R1, max_errors=0.1, min_overlap=3)
    out = {"reads": n, "fastq_bytes": size, "record_bytes": 317, "host_cpus": os.cpu_count(),
           "usable_cpus": len(os.sched_getaffinity(0)), "file": path, "runs": [],
           "passes_over_the_file": REPEAT,
           "what": "k feeders on devices=[0]*k; stub = no device work (host side of a feeder only); threads per feeder = 3"}

    def timed(label, **kw):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            r = trim_fastq_gpu(path, None, [adapter], threads=3, _repeat=REPEAT, **kw)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        k = len(kw.get("devices", [0]))
        row = {"what": label, "seconds": best, "Mreads_per_s": n * REPEAT / best / 1e6, "GB_per_s_in": size * REPEAT / best / 1e9,
               "per_feeder_Mreads_per_s": n * REPEAT / best / 1e6 / k, "way": r["way"]}
        if kw.get("feeder") == "process":
            # inside the feeder processes (their chunk loops; the ~1.5 s it takes to start an interpreter, import torch
            # and open a HIP context -- paid once per job -- are in "seconds" only)
            inner = max(v for v in [r.get("wall_s")] if v)
            per = [pd for pd in r["per_device"].values()]
            row["feeder_GB_per_s_in"] = [round(pd["GB_per_s_in"], 2) for pd in per]
            row["steady_Mreads_per_s"] = sum(pd["bytes_in"] / 317 for pd in per) / max(
                max(pd["bytes_in"] / max(pd["GB_per_s_in"], 1e-9) / 1e9 for pd in per), 1e-9) / 1e6
        out["runs"].append(row)
        print(row, file=sys.stderr)

    trim_fastq_gpu(path, None, [adapter], threads=3, devices=[0])                     # warm-up (plans, buffers, page cache)
    for k in (1, 2, 4, 8):
        timed(f"stub, {k} feeder(s), threads of one process", devices=[0] * k, _stub_device=True)
    for k in (1, 2, 4, 8):
        timed(f"stub, {k} feeder process(es)", devices=[0] * k, _stub_device=True, feeder="process")
    for k in (1, 2):
        timed(f"real pipeline, {k} feeder(s), threads of one process", devices=[0] * k)
        timed(f"real pipeline, {k} feeder process(es)", devices=[0] * k, feeder="process")
    os.remove(path)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":                                   # (the feeder processes are spawned: they import this file)
    main()
