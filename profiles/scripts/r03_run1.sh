#!/bin/bash
# round 3, GPU call 1: k_filter_stream2 correctness (both copy variants) + C2 A/B against the round-2 kernel
set -x
mkdir -p gpurun_out/r1
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q > gpurun_out/r1/test_stream_buf.log 2>&1; echo "rc=$?" >> gpurun_out/r1/test_stream_buf.log
CAH_S2_GLOBAL=1 timeout 900 python -m pytest tests/test_gpu_stream.py -x -q > gpurun_out/r1/test_stream_global.log 2>&1; echo "rc=$?" >> gpurun_out/r1/test_stream_global.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1/bench_s2_buf.json 2> gpurun_out/r1/bench_s2_buf.err
CAH_S2_GLOBAL=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1/bench_s2_global.json 2> gpurun_out/r1/bench_s2_global.err
CAH_NO_STREAM2=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1/bench_old.json 2> gpurun_out/r1/bench_old.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r1/prof" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --check-reads 0 > "$GRAFT_REPO_ROOT/gpurun_out/r1/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r1/prof.err"
cd "$GRAFT_REPO_ROOT"
find gpurun_out/r1/prof -name "*kernel_stats*" | head
tail -3 gpurun_out/r1/test_stream_buf.log gpurun_out/r1/test_stream_global.log
for f in bench_s2_buf bench_s2_global bench_old; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    j=json.loads(open(f"gpurun_out/r1/{f}.json").read().strip().splitlines()[-1])
    print(f, j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_step"], j["config"]["parity_check"][:40])
except Exception as e:
    print(f, "FAILED", e); print(open(f"gpurun_out/r1/{f}.err").read()[-1500:])
PY
done
