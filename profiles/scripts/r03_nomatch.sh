#!/bin/bash
# the prefilter with parts switched off (CAH_S2_NOMATCH: 1 copy only, 2 copy without result rows, 4 all but the rows)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/nm
for v in "" "CAH_S2_NOMATCH=1" "CAH_S2_NOMATCH=2" "CAH_S2_NOMATCH=4" "CAH_S2_NOMATCH=5" "CAH_S2_NOMATCH=3"; do
env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --check-reads 0 > gpurun_out/nm/b.json 2> gpurun_out/nm/b.err
python - "$v" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/nm/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1] or "full", {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()})
except Exception as e:
    print("FAILED", e, open("gpurun_out/nm/b.err").read()[-800:])
PY
done
