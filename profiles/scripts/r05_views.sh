#!/bin/bash
# one gpurun call: the end-aligned streaming of views inside a uniform batch (k_filter_stream2's RV form) -- parity tests,
# then bench.py on C2 uniform, --ragged (views) and, with PACKED=1, --ragged-packed (the per-lane kernels); C5 --ragged
out=gpurun_out/r05views; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_stream.py -q -m gpu -x > $out/tests.log 2>&1; tail -2 $out/tests.log
timeout 300 python bench.py --config C2 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 20000 > $out/c2_uniform.json 2> $out/c2_uniform.err
timeout 300 python bench.py --config C2 --ragged --steps 4 --warmup 1 --no-cpu-baseline > $out/c2_views.json 2> $out/c2_views.err
[ -n "$PACKED" ] && timeout 300 python bench.py --config C2 --ragged-packed --steps 3 --warmup 1 --no-cpu-baseline > $out/c2_packed.json 2> $out/c2_packed.err
timeout 300 python bench.py --config C5 --ragged --steps 3 --warmup 1 --no-cpu-baseline > $out/c5_views.json 2> $out/c5_views.err
python - <<'PY'
import json
for f in ("c2_uniform", "c2_views", "c2_packed", "c5_views"):
    try:
        r = json.loads(open(f"gpurun_out/r05views/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r["value"]), {k: round(v, 3) for k, v in r["roofline"]["kernel_ms_per_step"].items()}, r["config"]["parity_check"][:40])
    except Exception as e:
        print(f, "failed", e)
PY
