#!/bin/bash
# latency-side counters of the prefilter (accumulated in-flight levels / instruction counts = average latency)
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
envs="$1"; shift
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
run() { name="$1"; shift
    env $envs timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o p -- \
        python "$GRAFT_REPO_ROOT/bench.py" --config C2 --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > "$out/$name.json" 2> "$out/$name.err"; }
run l1 SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS SQ_LEVEL_WAVES
run l2 SQ_IFETCH_LEVEL SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC
cd "$GRAFT_REPO_ROOT"
python - "$out" <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
for sub in ("l1","l2"):
    for path in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if "filter_stream" in r["Kernel_Name"] and int(r["End_Timestamp"])-int(r["Start_Timestamp"])>1e6:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(sub, {k: f"{sum(v)/len(v):.4g}" for k,v in agg.items()})
PY
