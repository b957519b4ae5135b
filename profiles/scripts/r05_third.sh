#!/bin/bash
# round 5, third call: k_back_scan3 (scan3.hip) -- the scan's GPU tests, then A/B against k_back_scan (CAH_NO_SCAN3=1)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r05c; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_configs.py tests/test_gpu_small.py -x -q -m gpu --timeout 600 > $out/tests.log 2>&1; tail -n 25 $out/tests.log | cut -c1-400
ab() {  # tag env config steps extra...
  tag=$1; envs=$2; cfg=$3; steps=$4; shift 4
  env $envs timeout 400 python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-other-configs --check-reads 200000 "$@" > $out/b_$tag.json 2> $out/b_$tag.err
  python - "$tag" "$out/b_$tag" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]+".json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:30])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]+".err").read()[-800:])
PY
}
ab old_c2 CAH_NO_SCAN3=1 C2 10
ab new_c2 X=1 C2 10
ab old_c2_p1 CAH_NO_SCAN3=1 C2 5 --p-adapter 1
ab new_c2_p1 X=1 C2 5 --p-adapter 1
ab new_c2_p0 X=1 C2 5 --p-adapter 0
ab old_c3 CAH_NO_SCAN3=1 C3 5
ab new_c3 X=1 C3 5
ab old_c5 CAH_NO_SCAN3=1 C5 3
ab new_c5 X=1 C5 3
