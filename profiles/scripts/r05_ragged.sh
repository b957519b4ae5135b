#!/bin/bash
# ragged batches: C2 / C4 / C5 with the reads cut to 30 .. 150 characters, next to the uniform runs (one call)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r05rag; mkdir -p $out
run() {  # tag config steps extra
  tag=$1; cfg=$2; steps=$3; shift 3
  timeout 600 python bench.py --config $cfg --steps $steps --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 200000 "$@" > $out/b_$tag.json 2> $out/b_$tag.err
  python - "$tag" "$out/b_$tag" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]+".json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:30], "matched", round(j["config"]["matched_fraction"],4))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]+".err").read()[-1200:])
PY
}
run c2_uniform C2 5
run c2_ragged C2 5 --ragged
run c4_uniform C4 3
run c4_ragged C4 3 --ragged
run c5_ragged C5 3 --ragged
