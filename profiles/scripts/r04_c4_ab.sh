#!/bin/bash
# A/B inside one call: the library against developer builds of it (CAH_LIB_PATH), C4, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04ab
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 200000 > gpurun_out/r04ab/v.json 2> gpurun_out/r04ab/v.err
  python - "$tag" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/r04ab/v.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "|", round(j["value"],1), "Mreads/s", round(j["ms_per_step"],2), "ms", {k: round(v,2) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:3])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/r04ab/v.err").read()[-600:])
PY
}
for v in "$@"; do
  if [ "$v" = "base" ]; then run base X=1; else run $v CAH_LIB_PATH=$GRAFT_REPO_ROOT/cutadapt_amd/libcutadapt_hip_$v.so; fi
done
