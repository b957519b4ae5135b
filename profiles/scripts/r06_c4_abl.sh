#!/bin/bash
# developer builds of k_multi_stream (-DM2_ABL=2: events dropped, 3: ... and no tail probes, 7: ... and no events from the main
# pass) against the product build: where the prefilter's time goes (results of the ablated builds are wrong by design)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r06abl; mkdir -p $out
for v in prod abl2 abl3 abl7; do
  lib=$PWD/gpurun_in/lib_$v.so; [ "$v" = "prod" ] && lib=$PWD/cutadapt_amd/libcutadapt_hip.so
  CAH_LIB_ANY_ABI=1 CAH_LIB_PATH=$lib timeout 600 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 0 > $out/b_$v.json 2> $out/b_$v.err
  python - "$v" "$out/b_$v" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]+".json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]+".err").read()[-800:])
PY
done
