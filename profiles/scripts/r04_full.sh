#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04full
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04full/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r04full/gpu_tests.log
timeout 900 python bench.py > gpurun_out/r04full/bench_default.json 2> gpurun_out/r04full/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r04full/bench_default.json").read().strip().splitlines()[-1])
print("C2", round(j["value"],1), j["unit"], round(j["ms_per_step"],2), "ms", {k: round(v,2) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, "frac", round(j["roofline"]["frac"],3), j["config"]["parity_check"][:2])
for c,o in j.get("other_configs",{}).items():
    print(c, round(o.get("value",0),1), round(o.get("ms_per_step",0),2), "ms", o.get("parity_check","")[:2], o.get("error",""))
for c,o in j.get("p_adapter_extremes",{}).items():
    print(c, round(o.get("value",0),1), round(o.get("ms_per_step",0),2), "ms", o.get("parity_check","")[:2], o.get("error",""))
cb=j.get("cpu_baseline",{}); print("cpu", cb.get("value"), cb.get("cores"), j.get("gpu_over_cpu"))
PY
