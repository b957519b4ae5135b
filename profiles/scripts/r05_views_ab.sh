#!/bin/bash
# one gpurun call: RV form, next piece's views fetched early (the library) against fetched when needed (-DS2_RV_EARLY=0)
out=gpurun_out/r05views; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_stream.py -q -m gpu -x > $out/tests_early.log 2>&1; tail -2 $out/tests_early.log
for rep in 1 2; do
for v in early late; do
  lib=""; [ $v = late ] && lib="CAH_LIB_PATH=$PWD/cutadapt_amd/libcutadapt_hip_rvlate.so"
  env $lib timeout 300 python bench.py --config C2 --ragged --steps 4 --warmup 1 --no-cpu-baseline --check-reads 20000 > $out/c2_views_$v$rep.json 2> $out/c2_views_$v$rep.err
done; done
timeout 300 python bench.py --config C2 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 20000 > $out/c2_uniform.json 2> $out/c2_uniform.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05views/c2_views_*.json")) + ["gpurun_out/r05views/c2_uniform.json"]:
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(r["value"]), {k: round(v, 3) for k, v in r["roofline"]["kernel_ms_per_step"].items()}, r["config"]["parity_check"][:30])
    except Exception as e:
        print(f, "failed", e)
PY
