#!/bin/bash
# the round-end checks: whole -m gpu suite, smoke(), the default bench line (C2 + other_configs + CPU baselines)
out=gpurun_out/${1:-full}
mkdir -p $out
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
tail -n 5 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -n 2 $out/smoke.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python - "$out" <<'PY'
import json,sys
out=sys.argv[1]
try:
    j=json.loads(open(f"{out}/bench_default.json").read().strip().splitlines()[-1])
    print("C2", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, "frac", round(j["roofline"]["frac"],3), j["roofline"]["profile"], "cpu", j.get("cpu_baseline",{}).get("value"))
    for c,o in j.get("other_configs",{}).items():
        print(c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:40], o["roofline"]["kernel"], round(o["roofline"]["frac"],4), (o.get("cpu_baseline") or {}).get("value")))
except Exception as e:
    print("FAILED", e); print(open(f"{out}/bench_default.err").read()[-2000:])
PY
