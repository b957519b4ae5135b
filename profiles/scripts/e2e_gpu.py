#!/usr/bin/env python3
"""End to end, host-fed: FASTQ bytes in host memory -> trimmed FASTQ bytes in host memory, one MI355X.
Records are parsed, matched and formatted on the GPU (cutadapt_amd/gpu_pipeline.py); the host only cuts the
input at record starts and moves bytes.  Compared with the host-side batch pipeline (pipeline.trim_fastq,
parse/format on CPU threads).
Usage: python profiles/scripts/e2e_gpu.py [n_reads] [--devices all|0,1,..] [--threads T] > profiles/rNN/e2e_gpu.json
--devices: one feeder per named GPU (T worker threads each); the per-device chunk counts, input rates and busy
fractions of every run are part of the JSON and printed to stderr."""
import io
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
from cutadapt_amd.pipeline import trim_fastq

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("n_reads", nargs="?", type=int, default=30_000_000)
ap.add_argument("--devices", default=None, help="'all' or a comma-separated list of device indices")
ap.add_argument("--threads", type=int, default=4, help="worker threads per device in the --devices runs")
ap.add_argument("--file", default=None, help="also time a run that reads the FASTQ from this path (it is written first)")
args = ap.parse_args()
n = args.n_reads
devices = None if args.devices is None else ("all" if args.devices == "all" else [int(x) for x in args.devices.split(",")])
dev = torch.device("cuda", 0)
batch = workloads.device_batch("C2", n, device=dev)
seqs = batch.seqs.view(n, 150)
idx = torch.arange(n, dtype=torch.int64, device=dev)
name = torch.empty((n, 13), dtype=torch.uint8, device=dev)
name[:, 0] = ord("@"); name[:, 1] = ord("r"); name[:, 12] = 10
for k in range(10):
    name[:, 11 - k] = ((idx // (10 ** k)) % 10 + 48).to(torch.uint8)
rec = torch.empty((n, 317), dtype=torch.uint8, device=dev)
rec[:, :13] = name
rec[:, 13:163] = seqs
rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
rec[:, 166:316] = ord("I")
rec[:, 316] = 10
fastq = torch.empty(n * 317, dtype=torch.uint8).pin_memory()
fastq.copy_(rec.view(-1))
del rec, name, idx, batch, seqs
torch.cuda.empty_cache()
adapter = BackAdapter(workloads.TRUSEQ_R1, max_errors=0.1, min_overlap=3)
out = {"reads": n, "fastq_bytes": int(fastq.numel()), "record_bytes": 317, "runs": []}


def timed(label, fn, reps=2):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    n_run = int(stats.get("reads", n))                       # (a run over a part of the data is rated on that part)
    out["runs"].append({"what": label, "reads": n_run, "seconds": best, "Mreads_per_s": n_run / best / 1e6,
                        "GB_per_s_in": n_run * 317 / best / 1e9, "with_adapters": stats["with_adapters"],
                        "reverse_complemented": stats.get("reverse_complemented"), "way": stats.get("way"),
                        "bytes_out": stats.get("bytes_out"), "devices_used": stats.get("devices_used"),
                        "per_device": stats.get("per_device")})
    print(out["runs"][-1], file=sys.stderr)
    return stats


trim_fastq_gpu(fastq[: 317 * 200_000], None, [adapter], threads=2)          # warm-up
if devices is not None:
    # the multi-GPU mode: one feeder per device, chunks dealt round-robin, ordered merge; nothing else is timed
    trim_fastq_gpu(fastq[: 317 * 2_000_000], None, [adapter], threads=args.threads, devices=devices)   # every worker's buffers exist
    for assemble in ("device", "mixed"):
        timed(f"GPU parse+match+format, pinned input, no sink, devices={args.devices}, {args.threads} worker threads per device, assemble={assemble}",
              lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, assemble=assemble))
    timed(f"all-device way with -q 0,10 -m 20 in front of / behind the adapter step, devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, quality_cutoff=(0, 10), minimum_length=20))
    timed(f"all-device way with -q 0,10 in front of and --poly-a --max-ee 5 -m 20 behind the adapter step, devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, quality_cutoff=(0, 10), poly_a=True,
                                 max_expected_errors=5.0, minimum_length=20))
    timed(f"all-device way with -q 0,10 --times 2, devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, times=2, quality_cutoff=(0, 10)))
    from cutadapt_amd.adapters import FrontAdapter, LinkedAdapter
    linked = LinkedAdapter(FrontAdapter("ACGTACGTTT"), BackAdapter(workloads.TRUSEQ_R1, max_errors=0.1, min_overlap=3), False, True, "linked")
    timed(f"all-device way with one linked adapter (optional 5' part ... required 3' TruSeq) and -q 0,10 -m 20, devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [linked], threads=args.threads, devices=devices, quality_cutoff=(0, 10), minimum_length=20))
    linked2 = LinkedAdapter(FrontAdapter("TTGACCAGTA"), BackAdapter(workloads.TRUSEQ_R1[:20], max_errors=0.1, min_overlap=3), False, False, "linked2")
    timed(f"two linked adapters and -q 0,10 -m 20 (round 6: all-device way; rounds 3-5: the general way), devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [linked, linked2], threads=args.threads, devices=devices, quality_cutoff=(0, 10), minimum_length=20))
    timed(f"--times 2 --action mask (round 6: all-device way, the reads marked in place on the device; rounds 3-5: the general way), devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, times=2, action="mask"))
    timed(f"-q 0,10 --action mask (round 6: all-device way), devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, quality_cutoff=(0, 10),
                                 action="mask"), reps=1)
    timed(f"--action mask --poly-a (all-device way: the reads are marked in place, the poly-A trimmer sees the marked read), devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, action="mask", poly_a=True), reps=1)
    timed(f"--revcomp (round 6: all-device way -- both orientations matched, the better one turned around in place in HBM; rounds 3-5: the general way), devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, revcomp=True))
    # ... on data where every other read is the OTHER strand (the case --revcomp exists for): turned on the device here
    half = torch.empty(fastq.numel(), dtype=torch.uint8).pin_memory()
    comp = torch.arange(256, dtype=torch.uint8)
    for a, b in ("AT", "TA", "CG", "GC"):
        comp[ord(a)] = ord(b)
    for lo in range(0, n, 4_000_000):
        hi = min(n, lo + 4_000_000)
        part = fastq[lo * 317: hi * 317].to(dev).view(hi - lo, 317)
        odd = part[1::2, 13:163]
        part[1::2, 13:163] = comp.to(dev)[odd.flip(1).long()]
        half[lo * 317: hi * 317].copy_(part.view(-1))
    del part, odd
    timed(f"--revcomp -q 0,10 -m 20, every other read given as its reverse complement, devices={args.devices}",
          lambda: trim_fastq_gpu(half, None, [adapter], threads=args.threads, devices=devices, revcomp=True, quality_cutoff=(0, 10),
                                 minimum_length=20))
    devnull = open(os.devnull, "wb")
    timed(f"--info-file (round 6: all-device way, the rows formatted on the device: twice the bytes come back), devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, info_file=devnull))
    timed(f"--revcomp --info-file -q 0,10 -m 20, every other read given as its reverse complement (all-device way), devices={args.devices}",
          lambda: trim_fastq_gpu(half, None, [adapter], threads=args.threads, devices=devices, revcomp=True, quality_cutoff=(0, 10),
                                 minimum_length=20, info_file=devnull), reps=1)
    timed(f"--times 2 --info-file (round 6: all-device way, every round's rows formatted on the device), devices={args.devices}",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=args.threads, devices=devices, times=2, info_file=devnull), reps=1)
    timed(f"the general way, forced (--times 2 --info-file: the rows written by the host from the device's record index), devices={args.devices}",
          lambda: trim_fastq_gpu(fastq[: 317 * 4_000_000], None, [adapter], threads=args.threads, devices=devices, times=2,
                                 info_file=devnull, _general=True), reps=1)
    timed(f"the general way (--revcomp --times 2: both strands matched, the records of the better one written by the host), devices={args.devices}",
          lambda: trim_fastq_gpu(half[: 317 * 4_000_000], None, [adapter], threads=args.threads, devices=devices, revcomp=True,
                                 times=2), reps=1)
    del half
    if args.file:
        with open(args.file, "wb") as f:
            f.write(memoryview(fastq.numpy()))
        timed(f"the same from a file (page cache): byte ranges read by the feeders, devices={args.devices}",
              lambda: trim_fastq_gpu(args.file, None, [adapter], threads=args.threads, devices=devices))
        os.unlink(args.file)
    print(json.dumps(out, indent=1))
    sys.exit(0)
for threads, chunk_mib in ((1, 64), (2, 64), (3, 64), (4, 64), (6, 64), (8, 64), (4, 32), (4, 128)):
    timed(f"GPU parse+match+format, pinned input, no sink, {threads} worker thread(s) x {chunk_mib} MiB chunks",
          lambda: trim_fastq_gpu(fastq, None, [adapter], threads=threads, chunk_bytes=chunk_mib << 20))
devnull = open(os.devnull, "wb")
timed("GPU parse+match+format, pinned input, output written to /dev/null, 4 worker threads",
      lambda: trim_fastq_gpu(fastq, devnull, [adapter], threads=4))
pageable = fastq.numpy().copy()
timed("GPU parse+match+format, pageable numpy input (staged through pinned buffers), no sink, 4 worker threads",
      lambda: trim_fastq_gpu(pageable, None, [adapter], threads=4), reps=1)
del pageable
sub = fastq[: 317 * min(n, 4_000_000)].numpy().tobytes()
n_sub = min(n, 4_000_000)
t0 = time.perf_counter()
st = trim_fastq(io.BytesIO(sub), devnull, [adapter], threads=16)
dt = time.perf_counter() - t0
out["runs"].append({"what": "host-side batch pipeline (parse/format on 16 CPU threads), same data, /dev/null",
                    "seconds": dt, "Mreads_per_s": n_sub / dt / 1e6, "with_adapters_fraction": st["with_adapters"] / n_sub})
print(json.dumps(out, indent=1))
