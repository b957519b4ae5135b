#!/bin/bash
# last call of round 5 (final sources): the seeded parity suites with one more seed offset, then the driver-form line once more
# (another box: the spread between boxes is +- 3 %)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05last; mkdir -p $out
CAH_TEST_SEED_OFFSET=808 timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_multi2.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_small.py tests/test_gpu_long.py -q -m gpu --timeout 600 2>&1 | tail -n 3 > $out/seed_808.log
echo "seed offset 808: $(tail -n 1 $out/seed_808.log)"
timeout 900 python bench.py > $out/bench_default_2.json 2> $out/bench_default_2.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05last/bench_default_2.json").read().strip().splitlines()[-1])
print("C2", round(j["value"]), round(j["ms_per_step"], 3), {k: round(v, 3) for k, v in j["roofline"]["kernel_ms_per_step"].items()}, "frac", round(j["roofline"]["frac"], 3), "traffic", j["roofline"]["traffic"])
for c, o in j.get("other_configs", {}).items():
    print(c, o.get("error") or (round(o["value"]), round(o["ms_per_step"], 2), o["parity_check"][:30]))
for c, o in j.get("p_adapter_extremes", {}).items():
    print(c, o.get("error") or (round(o["value"]), o["parity_check"][:20]))
for c, o in j.get("ragged", {}).items():
    print("ragged", c, o.get("error") or (round(o["value"]), o["parity_check"][:20], round(o["vs_uniform"], 3)))
PY
