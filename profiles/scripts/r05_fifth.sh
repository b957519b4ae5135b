#!/bin/bash
# round 5, fifth call: product against round 4's library after the scalar thresholds + interior loads; the whole GPU suite
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r05f; mkdir -p $out
ab() {  # tag lib config steps extra
  tag=$1; lib=$2; cfg=$3; steps=$4; shift 4
  CAH_LIB_ANY_ABI=1 CAH_LIB_PATH=$lib timeout 400 python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-other-configs --check-reads 200000 "$@" > $out/b_$tag.json 2> $out/b_$tag.err
  python - "$tag" "$out/b_$tag" <<'PY'
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
ab base_c2b $P/libcutadapt_hip_base.so C2 10
ab prod_c2b $P/libcutadapt_hip.so C2 10
ab prod_c2_p1 $P/libcutadapt_hip.so C2 5 --p-adapter 1
ab prod_c3 $P/libcutadapt_hip.so C3 5
ab prod_c4 $P/libcutadapt_hip.so C4 3
ab prod_c5 $P/libcutadapt_hip.so C5 3
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 600 > $out/tests.log 2>&1; tail -n 5 $out/tests.log | cut -c1-300
