#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/dbg
for v in "" nolds nostep neither; do
lib=$PWD/cutadapt_amd/libcutadapt_hip${v:+_$v}.so
CAH_LIB_PATH=$lib timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check-reads 0 > gpurun_out/dbg/b.json 2> gpurun_out/dbg/b.err
python - "$v" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/dbg/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1] or "product", {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["prefilter_pass_fraction"])
except Exception as e:
    print("FAILED", e, open("gpurun_out/dbg/b.err").read()[-800:])
PY
done
