#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r3 gpurun_out/t
timeout 40 python -m pytest tests/test_gpu_fastq_general.py -x -q -m gpu -k "pair" > gpurun_out/t/pairs.log 2>&1; echo "tests rc=$?"; tail -n 3 gpurun_out/t/pairs.log
timeout 40 python profiles/scripts/r03_e2e_paired_files.py 4000000 1 4 8 2>&1 | tail -n 6
