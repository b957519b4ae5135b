#!/bin/bash
# quick loop for k_back_scan3: parity on a sample, the trace, C2 old/new
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r05d; mkdir -p $out
CAH_SCAN3=1 python gpurun_in/s3_dbg.py 200000 2>&1 | grep -v "amdgpu.ids\|Exception ignored\|Traceback\|oracle.py\|TypeError" | cut -c1-300
CAH_SCAN3=1 CAH_LIB_PATH=$PWD/cutadapt_amd/libcutadapt_hip_trace.so python profiles/scripts/scan3_trace.py 2>&1 | grep -v amdgpu.ids | head -9
ab() {
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
for rep in 1 2; do
ab old_c2 X=1 C2 10
ab new_c2 CAH_SCAN3=1 C2 10
ab new5_c2 "CAH_SCAN3=1 CAH_LIB_PATH=$PWD/cutadapt_amd/libcutadapt_hip_s3w5.so" C2 10
done
ab old_c2_p1 X=1 C2 5 --p-adapter 1
ab new_c2_p1 CAH_SCAN3=1 C2 5 --p-adapter 1
