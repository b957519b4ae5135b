#!/bin/bash
# end of round 3: host-fed figures with --times 2 and a linked adapter on the all-device way, then the round-end checks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 200 python profiles/scripts/e2e_gpu.py 30000000 --devices 0 --threads 8 > gpurun_out/r3/e2e_times.json 2> gpurun_out/r3/e2e_times.err; echo "e2e rc=$?"
grep -o "Mreads_per_s.: [0-9.]*\|what.: .[^,]*" gpurun_out/r3/e2e_times.err | paste - - | cut -c1-160
bash profiles/scripts/r03_full.sh final2
