#!/bin/bash
# round 6, first call: the new scale / multi-rank tests, then the whole -m gpu suite, smoke(), the default bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06first; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_multi2.py::test_pool_overflow_fails_loudly tests/test_gpu_fastq_general.py::test_every_visible_gpu_is_fed -x -q -m gpu --timeout 900 > $out/new_tests.log 2>&1; echo "rc=$?" >> $out/new_tests.log
tail -n 30 $out/new_tests.log
timeout 1200 python -m pytest tests -x -q -m gpu --timeout 900 --deselect tests/test_gpu_scale.py > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
tail -n 6 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python - "$out" <<'PY'
import json,sys
out=sys.argv[1]
try:
    j=json.loads(open(f"{out}/bench_default.json").read().strip().splitlines()[-1])
    print("C2", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, "frac", round(j["roofline"]["frac"],3), j["roofline"]["profile"], "cpu", j.get("cpu_baseline",{}).get("value"))
    print(j["config"]["parity_check"])
    for c,o in j.get("other_configs",{}).items():
        print(c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"], o["roofline"]["kernel"], round(o["roofline"]["frac"],4), (o.get("cpu_baseline") or {}).get("value")))
    for c,o in j.get("p_adapter_extremes",{}).items():
        print(c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:200]))
    for c,o in j.get("ragged",{}).items():
        print("ragged", c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:200], "x uniform", round(o["vs_uniform"],3)))
except Exception as e:
    print("FAILED", e); print(open(f"{out}/bench_default.err").read()[-2000:])
PY
