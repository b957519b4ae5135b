#!/bin/bash
# quick GPU check of the prefilter: stream tests + C2 bench lines (default / CAH_S2_GLOBAL / round-2 kernel)
out=gpurun_out/${1:-q}
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q > $out/test_stream.log 2>&1; echo "rc=$?" >> $out/test_stream.log
tail -n 4 $out/test_stream.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_s2.json 2> $out/bench_s2.err
if [ -n "$2" ]; then
CAH_S2_GLOBAL=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_s2_global.json 2> $out/bench_s2_global.err
CAH_NO_STREAM2=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_old.json 2> $out/bench_old.err
fi
for f in bench_s2 bench_s2_global bench_old; do [ -f $out/$f.json ] && python - "$out/$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    j=json.loads(open(f"{f}.json").read().strip().splitlines()[-1])
    print(f, round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:30])
except Exception as e:
    print(f, "FAILED", e); print(open(f"{f}.err").read()[-1500:])
PY
done
