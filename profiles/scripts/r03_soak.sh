#!/bin/bash
# soak: the seeded GPU parity tests with shifted seeds (tests/conftest.py: CAH_TEST_SEED_OFFSET); logs -> gpurun_out/soak/
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/soak; export TMPDIR=/tmp
for off in "$@"; do
  CAH_TEST_SEED_OFFSET=$off timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_scan.py tests/test_gpu_small.py tests/test_gpu_long.py tests/test_gpu_multi.py tests/test_revcomp.py tests/test_gpu_fastq_general.py tests/test_gpu_fastq_device.py -q -m gpu -p no:cacheprovider > gpurun_out/soak/gpu_offset_$off.log 2>&1
  echo "offset $off rc=$?" | tee -a gpurun_out/soak/gpu_offset_$off.log
  tail -n 3 gpurun_out/soak/gpu_offset_$off.log
done
