#!/bin/bash
# round 6: the streaming multi-adapter path with sub-classed tail k-mers: parity suites, then C4 A/B against round 5's library (gpurun_in/lib_r05.so) in one call
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r06c4; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_multi2.py tests/test_gpu_multi.py tests/test_gpu_configs.py -x -q -m gpu --timeout 900 > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -n 15 $out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.log
for rep in 1 2; do
for v in r05 prod; do
  lib=$PWD/gpurun_in/lib_r05.so; [ "$v" = "prod" ] && lib=$PWD/cutadapt_amd/libcutadapt_hip.so
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
