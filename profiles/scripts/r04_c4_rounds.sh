#!/bin/bash
# C4 with tiles drawn from one counter and rounds that end when the page pool runs low: pool sizes side by side
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04r
timeout 900 python -m pytest tests/test_gpu_multi2.py tests/test_gpu_multi.py -x -q 2>&1 | tail -3
run() {
  env "$@" timeout 600 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 200000 > gpurun_out/r04r/v.json 2> gpurun_out/r04r/v.err
  python - "$*" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/r04r/v.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "|", round(j["value"],1), "Mreads/s", round(j["ms_per_step"],2), "ms", {k: round(v,2) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["roofline"].get("launches_per_step"), j["config"]["parity_check"][:3])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/r04r/v.err").read()[-600:])
PY
}
run X=1
run CAH_MULTI_PAIR_CAP=536870912
run CAH_MULTI_PAIR_CAP=1073741824
run CAH_MULTI_PAIR_CAP=4294967296
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04r/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --config C4 --steps 2 --warmup 0 --no-cpu-baseline --no-other-configs --check-reads 0 > /dev/null 2>&1
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04r/trace/**/*kernel_trace.csv",recursive=True)
rows=sorted(csv.DictReader(open(f[0])), key=lambda r:int(r["Start_Timestamp"]))
prev=None
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
    if not n.startswith("k_"): continue
    st,en=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if n.startswith(("k_multi","k_dp")): print("%-40s %9.3f ms  gap %8.3f ms"%(n,(en-st)/1e6,(st-prev)/1e6 if prev else 0))
    prev=en
PY
