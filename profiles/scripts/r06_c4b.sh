#!/bin/bash
# round 6, second C4 pass (exact bitmaps of the tail classes' short k-mers): parity suites, then C4 A/B against the library of
# the commit before (gpurun_in/lib_r06a.so) in one call.  $1 = tag, $2 = "notests" to skip the suites
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/${1:-r06c4b}; mkdir -p $out
if [ "$2" != "notests" ]; then
timeout 1500 python -m pytest tests/test_gpu_multi2.py tests/test_gpu_multi.py tests/test_gpu_configs.py -x -q -m gpu --timeout 900 > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -n 6 $out/tests.log
fi
for rep in 1 2; do
for v in r06a prod; do
  lib=$PWD/gpurun_in/lib_r06a.so; [ "$v" = "prod" ] && lib=$PWD/cutadapt_amd/libcutadapt_hip.so
  CAH_LIB_ANY_ABI=1 CAH_LIB_PATH=$lib timeout 600 python bench.py --config C4 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 250000 > $out/b_$v.json 2> $out/b_$v.err
  python - "$v" "$out/b_$v" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]+".json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:12], "matched", round(j["config"]["matched_fraction"],5))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]+".err").read()[-1500:])
PY
done; done
