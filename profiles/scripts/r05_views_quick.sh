#!/bin/bash
# quick A/B of the RV form: stream tests, then C2 --ragged twice and uniform once (same box)
out=gpurun_out/r05views; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_stream.py -q -m gpu -x > $out/tests.log 2>&1; tail -1 $out/tests.log
for i in 1 2; do timeout 300 python bench.py --config C2 --ragged --steps 4 --warmup 1 --no-cpu-baseline --check-reads 20000 > $out/q_views$i.json 2> $out/q_views$i.err; done
timeout 300 python bench.py --config C2 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 20000 > $out/q_uniform.json 2> $out/q_uniform.err
python - <<'PY'
import json
for f in ("q_views1", "q_views2", "q_uniform"):
    try:
        r = json.loads(open(f"gpurun_out/r05views/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r["value"]), {k: round(v, 3) for k, v in r["roofline"]["kernel_ms_per_step"].items()}, r["config"]["parity_check"][:20])
    except Exception as e:
        print(f, "failed", e)
PY
