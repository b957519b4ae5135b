#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_fastq_general.py tests/test_gpu_fastq_device.py -x -q 2>&1 | tail -12
