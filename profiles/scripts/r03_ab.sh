#!/bin/bash
# A/B of kernel libraries inside ONE gpurun call (boxes differ by a few per cent): r03_ab.sh <suffix> <suffix> ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in "$@"; do
lib=$PWD/cutadapt_amd/libcutadapt_hip${v:+_$v}.so
[ "$v" = "product" ] && lib=$PWD/cutadapt_amd/libcutadapt_hip.so
CAH_LIB_PATH=$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --check-reads 0 > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err
python - "$v" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/ab/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/ab/b.err").read()[-500:])
PY
done; done
