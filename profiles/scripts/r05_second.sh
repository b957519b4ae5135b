#!/bin/bash
# round 5, second call: round 4's library on this box (base), the scan's stations (SCAN_TRACE builds), the GPU suite
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r05b; mkdir -p $out
ab() {  # tag lib config steps
  CAH_LIB_ANY_ABI=1 CAH_LIB_PATH=$2 timeout 400 python bench.py --config $3 --steps $4 --warmup 2 --no-cpu-baseline --no-other-configs --check-reads 200000 > $out/b_$1.json 2> $out/b_$1.err
  python - "$1" "$out/b_$1" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]+".json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:30])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]+".err").read()[-800:])
PY
}
P=$PWD/cutadapt_amd
ab base_c2 $P/libcutadapt_hip_base.so C2 10
ab prod_c2 $P/libcutadapt_hip.so C2 10
ab plain_c2 $P/libcutadapt_hip_plain.so C2 10
ab base_c4 $P/libcutadapt_hip_base.so C4 3
ab base_c5 $P/libcutadapt_hip_base.so C5 3
for t in trace_base trace; do
  echo "== scan trace: $t"; CAH_LIB_ANY_ABI=1 CAH_LIB_PATH=$P/libcutadapt_hip_$t.so timeout 300 python profiles/scripts/scan_trace.py 2>&1 | tail -n 8
done > $out/scan_trace.txt 2>&1
cat $out/scan_trace.txt
timeout 1200 python -m pytest tests -x -q -m gpu --timeout 600 > $out/tests.log 2>&1; tail -n 5 $out/tests.log
