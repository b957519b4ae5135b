#!/bin/bash
# result rows cleared by the prefilter (default) or by memsets (CAH_NO_FILTER_CLEAR=1): whole-step time, same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/nm
for rep in 1 2; do for v in "" "CAH_NO_FILTER_CLEAR=1"; do
env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --check-reads 0 > gpurun_out/nm/c.json 2> gpurun_out/nm/c.err
python - "$v" <<'PY'
import json,sys
j=json.loads(open("gpurun_out/nm/c.json").read().strip().splitlines()[-1])
print(sys.argv[1] or "in-kernel", round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()})
PY
done; done
