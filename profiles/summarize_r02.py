#!/usr/bin/env python3
"""Condense rocprofv3 output (<src>/{trace,fetch,write,sq}) into the small files kept under profiles/:
<tag>_kernel_stats.csv, <tag>_pmc_summary.json and the entry of profiles/pmc_latest.json that bench.py
reads for roofline.traffic / roofline.valu.
Usage: python profiles/summarize_r02.py <src dir> <tag> <config> <reads per gpu>"""
import collections
import csv
import json
import os
import sys

src, tag, config, reads = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
here = os.path.dirname(os.path.abspath(__file__))
outdir = os.path.join(here, "r02")
os.makedirs(outdir, exist_ok=True)


def family(name):
    n = name.replace("void ", "")
    if n.startswith("k_filter") or n.startswith("k_multi_filter"):
        return "k_filter"
    if n.startswith("k_back_scan"):
        return "k_back_scan"
    if n.startswith("k_dp"):
        return "k_dp"
    if n.startswith("k_comparer"):
        return "k_comparer"
    return None


stats = f"{src}/trace/t_kernel_stats.csv"
if os.path.exists(stats):
    rows = list(csv.DictReader(open(stats)))
    with open(f"{outdir}/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if family(k) is None:
            continue
        k = k.split("(")[0].replace("void ", "")
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if dur < 0.05:                      # the variant of a kernel pair that returns at once
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_ms"].append(dur)
        agg[k]["_res"] = {x: int(r[x]) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                 "Scratch_Size", "LDS_Block_Size", "Grid_Size")}
    return agg


out = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python bench.py --config " + config +
                  " --no-cpu-baseline --check-reads 0 --steps 1 --warmup 0   (one pass per counter group)",
       "config": config, "reads_per_gpu": reads,
       "notes": ["FETCH_SIZE / WRITE_SIZE are KiB as printed by rocprofv3; per MI355X_MICROARCH.md FETCH_SIZE on gfx950 "
                 "under-reports wide coalesced streaming reads by 2x (uncalibrated for the per-lane unaligned 16-byte "
                 "loads used here), WRITE_SIZE is uncalibrated",
                 "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; per-launch averages"],
       "kernels": {}}
for f in ("fetch/f", "write/w", "sq/s"):
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
                o["launches_seen"] = len(v)
            else:
                o[c] = sum(v) / len(v)
json.dump(out, open(f"{outdir}/{tag}_pmc_summary.json", "w"), indent=1)

latest_path = os.path.join(here, "pmc_latest.json")
latest = json.load(open(latest_path)) if os.path.exists(latest_path) else {}
entry = {"reads_per_gpu": reads, "source": f"profiles/r02/{tag}_pmc_summary.json", "kernels": {}}
for k, d in out["kernels"].items():
    fam = family(k)
    e = entry["kernels"].setdefault(fam, {"kernel_full_name": k})
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # guide: hbm bytes = (FETCH_SIZE [x2 on gfx950] + WRITE_SIZE) * 1024
        e["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        e["hbm_bytes_per_launch_uncorrected"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if "SQ_INSTS_VALU" in d:
        e["valu_insts_per_launch"] = d["SQ_INSTS_VALU"]
        e["cycles_per_valu_inst"] = 3.0
latest[config] = entry
json.dump(latest, open(latest_path, "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, {c: (round(v, 1) if isinstance(v, float) else v) for c, v in d.items()})
