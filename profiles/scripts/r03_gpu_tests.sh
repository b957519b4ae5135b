#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/t; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu "$@" > gpurun_out/t/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/t/gpu_tests.log
tail -n 6 gpurun_out/t/gpu_tests.log
