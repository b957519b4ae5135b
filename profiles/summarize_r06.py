#!/usr/bin/env python3
"""Condense the output of profiles/scripts/pmc.sh (gpurun_out/<dir>/{trace,sq1,sq2,sq3,fetch,write,grbm}) into
profiles/r06/<tag>_kernel_stats.csv and profiles/r06/<tag>_pmc_summary.json (round 5: + the kernels' register / spill /
scratch figures read from the library BINARY, profiles/scripts/kernel_resources.py, and the library's cah_build_id).
Round 6: every launch of every k_* kernel is also SUMMED per profiled step (hbm_bytes_per_step per kernel and per family:
what bench.py reports as roofline.whole_step_traffic); --steps is the number of steps of the PMC passes (pmc.sh: 2).
Usage: python profiles/summarize_r06.py <src dir> <tag> [--min-ms 0.05] [--steps 2]"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
min_ms = float(sys.argv[sys.argv.index("--min-ms") + 1]) if "--min-ms" in sys.argv else 0.05
here = os.path.dirname(os.path.abspath(__file__))
outdir = os.path.join(here, "r06")
pmc_steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 2
os.makedirs(outdir, exist_ok=True)


def find(sub, pattern):
    hits = glob.glob(os.path.join(src, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


stats = find("trace", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(f"{outdir}/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])

out = {"command": "profiles/scripts/pmc.sh: rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python bench.py "
                  "... --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 (one pass per counter group; the kernel "
                  "stats come from a --kernel-trace --stats pass of 5 steps)",
       "notes": ["per-launch averages over the launches longer than %.2f ms" % min_ms,
                 "FETCH_SIZE / WRITE_SIZE are KiB as rocprofv3 prints them; MI355X_MICROARCH.md: FETCH_SIZE on gfx950 "
                 "reports half the bytes of wide coalesced streaming reads (hbm_fetch_bytes_x2 applies that)",
                 "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (4 shader cycles)"],
       "kernels": {}}
for sub in ("sq1", "sq2", "sq3", "fetch", "write", "grbm"):
    path = find(sub, "*counter_collection.csv")
    if not path:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    res = {}
    if sub in ("fetch", "write"):
        totals = out.setdefault("_totals", {}).setdefault(sub, collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_") or k.startswith("k_synth"):     # (k_synth: the workload generator, not part of a step)
            continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if sub in ("fetch", "write") and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            totals[k] += float(r["Counter_Value"])         # (every launch, short ones too: the step's sum)
        if dur < min_ms:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_ms_" + sub].append(dur)
        res[k] = {x: int(r[x]) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size",
                                         "LDS_Block_Size", "Grid_Size", "Workgroup_Size") if x in r}
    for k, d in agg.items():
        o = out["kernels"].setdefault(k, {})
        o["resources"] = res[k]
        for c, v in d.items():
            o[c] = sum(v) / len(v)
for k, o in out["kernels"].items():
    wc = o.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in o:
                o[c + "_frac_of_wave_cycles"] = o[c] / wc
    if "SQ_BUSY_CYCLES" in o and "SQ_WAVE_CYCLES" in o:
        o["waves_in_flight_avg_per_SE_busy_cycle"] = o["SQ_WAVE_CYCLES"] * 4 / o["SQ_BUSY_CYCLES"]
    if o.get("SQ_ACTIVE_INST_LDS") and "SQ_LDS_BANK_CONFLICT" in o:
        # cycles of conflict replays per quad-cycle the LDS pipe was active (both counters summed over the chip)
        o["lds_bank_conflict_cycles_per_lds_active_quadcycle"] = o["SQ_LDS_BANK_CONFLICT"] / o["SQ_ACTIVE_INST_LDS"]
        o["lds_bank_conflict_frac_of_lds_active_cycles"] = o["SQ_LDS_BANK_CONFLICT"] / (4.0 * o["SQ_ACTIVE_INST_LDS"])
    if "FETCH_SIZE" in o:
        o["hbm_fetch_bytes_x2"] = o["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in o:
        o["hbm_write_bytes"] = o["WRITE_SIZE"] * 1024
tot = out.pop("_totals", {})
per_step = {}
for k in set(tot.get("fetch", {})) | set(tot.get("write", {})):
    f, w = tot.get("fetch", {}).get(k, 0.0), tot.get("write", {}).get(k, 0.0)
    per_step[k] = {"hbm_fetch_bytes_x2_per_step": f * 1024 * 2 / pmc_steps, "hbm_write_bytes_per_step": w * 1024 / pmc_steps,
                   "hbm_bytes_per_step": (f * 1024 * 2 + w * 1024) / pmc_steps}
out["per_step"] = {"steps_profiled": pmc_steps, "kernels": per_step,
                   "hbm_bytes_per_step_all_kernels": sum(v["hbm_bytes_per_step"] for v in per_step.values())}
N_SIMD = 1024
for k, o in out["kernels"].items():
    # derived, all from counters of the same rocprofv3 passes (profiled clocks and durations, not the un-profiled run's)
    if "GRBM_GUI_ACTIVE" in o:
        cyc = o["GRBM_GUI_ACTIVE"] / 8.0                     # the counter sums the 8 XCDs
        o["kernel_cycles_profiled"] = cyc
        if "duration_ms_grbm" in o:
            o["clock_ghz_profiled"] = cyc / (o["duration_ms_grbm"] * 1e6)
        if "SQ_ACTIVE_INST_VALU" in o:
            o["valu_busy"] = o["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * cyc)
        if "SQ_INSTS_VALU" in o:
            # a wave64 VALU instruction occupies its SIMD for 2 cycles (and / or / xor / add / v_bitop3) or 4 (shifts, SDWA,
            # compares, selects, v_perm: profiles/r03/valu_ubench.txt): the busy fraction lies between these two
            o["valu_issue_lower"] = o["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * cyc)
            o["valu_issue_upper"] = min(1.0, o["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * cyc))
        if "SQ_WAVE_CYCLES" in o:
            o["waves_per_simd_time_average"] = o["SQ_WAVE_CYCLES"] * 4.0 / (N_SIMD * cyc)
    res = o.get("resources", {})
    if res.get("Workgroup_Size") and res.get("LDS_Block_Size") is not None:
        # resident waves per SIMD by the launch's resources (LDS: 160 KiB per CU; VGPR_Count is per lane as rocprofv3
        # prints it for wave64: half the allocation)
        wg_waves = res["Workgroup_Size"] // 64
        by_lds = (160 * 1024 // max(res["LDS_Block_Size"], 1)) if res["LDS_Block_Size"] else 99
        by_vgpr = 512 // max(8, ((2 * res.get("VGPR_Count", 64) + 7) // 8) * 8)
        o["waves_per_simd"] = min(8, by_vgpr, max(1, min(by_lds, 2048 // res["Workgroup_Size"]) * wg_waves // 4))

# round 5: registers, spills and scratch of every profiled kernel, from the code objects inside the library that ran
try:
    sys.path.insert(0, os.path.join(here, "scripts"))
    import kernel_resources
    lib = os.environ.get("CAH_LIB_PATH") or os.path.join(os.path.dirname(here), "cutadapt_amd", "libcutadapt_hip.so")
    kres = kernel_resources.kernels(lib)
    for k, o in out["kernels"].items():
        if k in kres:
            o["code_object"] = kres[k]
    sys.path.insert(0, os.path.dirname(here))
    from cutadapt_amd import build as _build
    out["library_build_id"] = _build.library_build_id(lib)
except Exception as exc:                                   # the counters must survive a tooling hiccup
    out["code_object_error"] = repr(exc)[:200]

if "--update-latest" in sys.argv:
    # the entry bench.py reads for roofline.traffic / roofline.valu, tied to the sources by their hash
    sys.path.insert(0, os.path.dirname(here))
    from bench import library_hash as csrc_hash            # (round 5: the hash the loaded BINARY carries, cah_build_id)
    config = sys.argv[sys.argv.index("--config") + 1]
    reads = int(sys.argv[sys.argv.index("--reads") + 1])
    latest_path = os.path.join(here, "pmc_latest.json")
    try:
        latest = json.load(open(latest_path))
    except Exception:
        latest = {}
    fam = {"k_filter": ("k_filter", "k_multi_filter", "k_multi_stream"), "k_back_scan": ("k_back_scan", "k_multi_scan"), "k_dp": ("k_dp",)}
    entry = {"reads_per_gpu": reads, "csrc_sha256": csrc_hash(), "source": f"profiles/r06/{tag}_pmc_summary.json", "kernels": {}}
    claimed = set()
    for name, prefixes in fam.items():
        cands = [(k, o) for k, o in out["kernels"].items() if k.startswith(prefixes)]
        if not cands:
            continue
        k, o = max(cands, key=lambda ko: ko[1].get("duration_ms_sq1", 0.0) * 1.0)
        members = [kk for kk in per_step if kk.startswith(prefixes)]
        claimed.update(members)
        entry["kernels"][name] = {
            "hbm_bytes_per_step": sum(per_step[kk]["hbm_bytes_per_step"] for kk in members) if members else None,
            "kernels_in_family": sorted(members),
            "kernel_full_name": k,
            "hbm_bytes_per_launch": (o.get("hbm_fetch_bytes_x2", 0.0) + o.get("hbm_write_bytes", 0.0)) or None,
            "hbm_fetch_bytes_per_launch_x2": o.get("hbm_fetch_bytes_x2"),
            "hbm_write_bytes_per_launch": o.get("hbm_write_bytes"),
            "valu_insts_per_launch": o.get("SQ_INSTS_VALU"),
            "salu_insts_per_launch": o.get("SQ_INSTS_SALU"),
            "lds_insts_per_launch": o.get("SQ_INSTS_LDS"),
            "valu_busy": o.get("valu_busy"),
            "valu_issue_lower": o.get("valu_issue_lower"),
            "valu_issue_upper": o.get("valu_issue_upper"),
            "waves_per_simd": o.get("waves_per_simd"),
            "wait_frac_of_wave_cycles": o.get("SQ_WAIT_ANY_frac_of_wave_cycles"),
            "issue_stall_frac_of_wave_cycles": o.get("SQ_WAIT_INST_ANY_frac_of_wave_cycles"),
            "clock_ghz_profiled": o.get("clock_ghz_profiled"),
            "profiled_ms_per_launch": o.get("duration_ms_sq1"),
        }
    rest = [kk for kk in per_step if kk not in claimed]
    if rest:      # decode / merge / clear kernels of the step that belong to no timed family
        entry["kernels"]["other"] = {"kernel_full_name": ", ".join(sorted(rest))[:300], "kernels_in_family": sorted(rest),
                                     "hbm_bytes_per_step": sum(per_step[kk]["hbm_bytes_per_step"] for kk in rest)}
    latest[config] = entry
    with open(latest_path, "w") as f:
        json.dump(latest, f, indent=1, sort_keys=True)

with open(f"{outdir}/{tag}_pmc_summary.json", "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out["kernels"], indent=1, sort_keys=True)[:6000])
