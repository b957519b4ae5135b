#!/bin/bash
# the LONG form of the streaming prefilter: tests, then 250 / 300 bp bench lines with and without it, and C2 (regression)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04long
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -4
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --check-reads 200000 $EXTRA > gpurun_out/r04long/$tag.json 2> gpurun_out/r04long/$tag.err
  python - "$tag" <<'PY'
import json,sys
try:
    j=json.loads(open(f"gpurun_out/r04long/{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "|", round(j["value"],1), "Mreads/s", round(j["ms_per_step"],2), "ms", {k: round(v,2) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, "dominant", j["roofline"]["kernel"], "frac", round(j["roofline"]["frac"],3), j["config"]["parity_check"][:2])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(f"gpurun_out/r04long/{sys.argv[1]}.err").read()[-600:])
PY
}
EXTRA="--read-len 250 --reads 60000000"; run len250_long X=1; run len250_lean CAH_NO_STREAM2_LONG=1
EXTRA="--read-len 300 --reads 50000000"; run len300_long X=1; run len300_lean CAH_NO_STREAM2_LONG=1
EXTRA=""; run c2 X=1
