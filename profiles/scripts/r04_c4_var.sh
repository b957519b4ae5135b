#!/bin/bash
# C4 bench variants inside one call: env assignments per variant
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_gpu_multi2.py -x -q 2>&1 | tail -2
run() {
  env "$@" timeout 600 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 200000 > gpurun_out/r04q/v.json 2> gpurun_out/r04q/v.err
  python - "$*" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/r04q/v.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "|", round(j["value"],1), "Mreads/s", round(j["ms_per_step"],2), "ms", {k: round(v,2) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:3])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/r04q/v.err").read()[-600:])
PY
}
run X=1
run CAH_MULTI_RESCAN=1
run CAH_MULTI_PAIR_CAP=1073741824
run CAH_MULTI_PAIR_CAP=4294967296
