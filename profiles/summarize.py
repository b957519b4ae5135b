#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/{trace,fetch,write,sq,sq2}) into the small files
kept under profiles/.  Usage: python profiles/summarize.py <tag> [reads_per_gpu]"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
src = "gpurun_out/prof"
here = os.path.dirname(os.path.abspath(__file__))

rows = list(csv.DictReader(open(f"{src}/trace/t_kernel_stats.csv")))
with open(f"{here}/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                    r["MinNs"], r["MaxNs"], r["StdDev"]])


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if k.startswith("void k_dp") or k.startswith("void k_filter"):
            k = k.split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["duration_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            agg[k]["_res"] = {x: int(r[x]) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                     "Scratch_Size", "LDS_Block_Size", "Grid_Size")}
    return agg


out = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline "
                  "--check-reads 0 --steps 1 --warmup 0   (one pass per counter group)",
       "reads_per_gpu": reads,
       "notes": ["FETCH_SIZE / WRITE_SIZE are KiB as printed by rocprofv3; per MI355X_MICROARCH.md FETCH_SIZE on gfx950 "
                 "under-reports wide coalesced streaming reads by 2x (uncalibrated for the per-lane unaligned 16-byte "
                 "loads used here), WRITE_SIZE is uncalibrated",
                 "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles"],
       "kernels": {}}
for f in ("fetch/f", "write/w", "sq/s", "sq2/s2"):
    path = f"{src}/{f}_counter_collection.csv"
    if not os.path.exists(path):
        continue
    for k, d in load(path).items():
        o = out["kernels"].setdefault(k, {})
        for c, v in d.items():
            if c == "_res":
                o["resources"] = v
            elif c == "duration_ms":
                o.setdefault("duration_ms", []).append(sum(v) / len(v))
            else:
                o[c] = sum(v) / len(v)
json.dump(out, open(f"{here}/{tag}_pmc_summary.json", "w"), indent=1)

# HBM traffic of the dominant kernel for bench.py's roofline.traffic
dom = max(out["kernels"].items(), key=lambda kv: sum(kv[1]["duration_ms"]) / len(kv[1]["duration_ms"]))
name, d = dom
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    traffic = {"kernel": "k_dp" if name.startswith("k_dp") else "k_filter", "kernel_full_name": name,
               "reads_per_gpu": reads,
               "fetch_KiB": d["FETCH_SIZE"], "write_KiB": d["WRITE_SIZE"],
               # guide: hbm_bytes = (FETCH_SIZE [x2 on gfx950 for wide streaming reads] + WRITE_SIZE) * 1024
               "bytes_per_launch": (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024,
               "bytes_per_launch_uncorrected": (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024,
               "source": f"profiles/{tag}_pmc_summary.json"}
    json.dump(traffic, open(f"{here}/hbm_traffic.json", "w"), indent=1)
    print(traffic)
for k, d in out["kernels"].items():
    print(k, {c: (round(v, 1) if isinstance(v, float) else v) for c, v in d.items()})
