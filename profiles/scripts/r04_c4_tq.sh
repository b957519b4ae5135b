#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_gpu_multi2.py -x -q 2>&1 | tail -3
bash profiles/scripts/r04_c4_quick.sh $1
