#!/bin/bash
# C4 bench (parity sample on) + optional PMC pass: r04_c4_quick.sh [tag]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04q
timeout 600 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 200000 > gpurun_out/r04q/bench_c4.json 2> gpurun_out/r04q/bench_c4.err
echo "bench rc $?"; python - <<'PY'
import json
try:
    j=json.loads(open("gpurun_out/r04q/bench_c4.json").read().strip().splitlines()[-1])
    print(round(j["value"],1), "Mreads/s", round(j["ms_per_step"],2), "ms", j["roofline"]["kernel_ms_per_step"], j["config"]["parity_check"])
except Exception as e:
    print("FAILED", e, open("gpurun_out/r04q/bench_c4.err").read()[-800:])
PY
[ -n "$1" ] && bash profiles/scripts/r04_pmc_c4.sh "$1" | cut -c1-900
