#!/bin/bash
# one gpurun call: SQ counters of k_filter_stream2 on C2, uniform reads against views (RV form) -- per launch averages
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r05views; mkdir -p $out
for mode in uniform views; do
  extra=""; [ $mode = views ] && extra="--ragged"
  o=/tmp/pmc_$mode; mkdir -p $o
  ( cd /tmp
    run() { name="$1"; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$o/$name" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --config C2 --no-other-configs $extra --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > "$o/$name.json" 2> "$o/$name.err"; }
    run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
    run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
    run sq3 SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_I8 )
done
python - <<'PY' > $out/pmc_compare.txt
import csv, glob, collections
res = {}
for mode in ("uniform", "views"):
    agg = collections.defaultdict(list)
    for sub in ("sq1", "sq2", "sq3"):
        for path in glob.glob(f"/tmp/pmc_{mode}/{sub}/**/*counter_collection.csv", recursive=True):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(path)):
                if "k_filter_stream2" in r["Kernel_Name"]:
                    per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (d, c), v in per.items():
                agg[c].append(v)
    res[mode] = {c: sum(v) / len(v) for c, v in agg.items()}
print(f"{'counter':28s} {'uniform':>16s} {'views':>16s} ratio")
for c in sorted(res["uniform"]):
    u, v = res["uniform"][c], res["views"].get(c, float('nan'))
    print(f"{c:28s} {u:16.0f} {v:16.0f} {v / u if u else float('nan'):.3f}")
PY
cat $out/pmc_compare.txt
