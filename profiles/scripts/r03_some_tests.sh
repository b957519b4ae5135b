#!/bin/bash
# run selected test files on the GPU box: r03_some_tests.sh tests/a.py tests/b.py ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/t; export TMPDIR=/tmp
timeout 2400 python -m pytest -x -q -m gpu "$@" > gpurun_out/t/some_tests.log 2>&1; echo "rc=$?" >> gpurun_out/t/some_tests.log
tail -n 60 gpurun_out/t/some_tests.log
