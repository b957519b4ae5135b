#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r06ragged; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_multi2.py tests/test_gpu_multi.py -x -q -m gpu --timeout 600 > $out/test.log 2>&1; echo "rc=$?" >> $out/test.log; tail -n 3 $out/test.log
for st in ${STREAMS:-1 4}; do
  CAH_BUCKET_STREAMS=$st timeout 900 python bench.py --config C4 --ragged --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 250000 > $out/b_$st.json 2> $out/b_$st.err
  python - "streams $st" "$out/b_$st" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]+".json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:12])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]+".err").read()[-1500:])
PY
done
