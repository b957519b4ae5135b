#!/bin/bash
# the straggler threshold of k_back_scan (CAH_SCAN_RETRY, default 12) with the cheaper column
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r05ab; mkdir -p $out
for rep in 1 2; do
for v in 12 6 16 20 28; do
  CAH_SCAN_RETRY=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --check-reads 0 > $out/b.json 2> $out/b.err
  python - "retry=$v" "$out" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]+"/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]+"/b.err").read()[-500:])
PY
done; done
