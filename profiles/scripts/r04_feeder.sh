#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04feed
timeout 900 python -m pytest tests/test_gpu_fastq_device.py -x -q -k "feeder_processes" 2>&1 | tail -5
timeout 1200 python profiles/scripts/r04_feeder_scaling.py 12000000 16 > gpurun_out/r04feed/feeder_scaling.json 2> gpurun_out/r04feed/feeder_scaling.err; echo "scaling rc $?"
tail -22 gpurun_out/r04feed/feeder_scaling.err | cut -c1-260
