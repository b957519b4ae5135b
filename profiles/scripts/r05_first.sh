#!/bin/bash
# round 5, first call: instruction costs (r05_valu_ubench), the cost scan's stations (SCAN_TRACE builds), A/B of the
# product library against round 4's scan (libcutadapt_hip_base.so) on C2 / C4, and the scan's GPU tests
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05a; mkdir -p $out
timeout 120 ./gpurun_in/r05_valu_ubench > $out/valu_ubench.txt 2>&1; tail -n 40 $out/valu_ubench.txt | cut -c1-110
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
for rep in 1 2; do
  ab base_c2_$rep $P/libcutadapt_hip_base.so C2 10
  ab prod_c2_$rep $P/libcutadapt_hip.so C2 10
done
ab base_c4 $P/libcutadapt_hip_base.so C4 3
ab prod_c4 $P/libcutadapt_hip.so C4 3
ab base_c5 $P/libcutadapt_hip_base.so C5 3
ab prod_c5 $P/libcutadapt_hip.so C5 3
for t in trace_base trace; do
  echo "== scan trace: $t"; CAH_LIB_ANY_ABI=1 CAH_LIB_PATH=$P/libcutadapt_hip_$t.so timeout 300 python profiles/scripts/scan_trace.py 2>&1 | tail -n 8
done > $out/scan_trace.txt 2>&1
cat $out/scan_trace.txt
timeout 1200 python -m pytest tests -x -q -m gpu --timeout 600 > $out/tests.log 2>&1; tail -n 5 $out/tests.log
