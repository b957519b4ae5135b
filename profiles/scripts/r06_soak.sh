#!/bin/bash
# the seeded parity suites with shifted seeds only (SOAK_SEEDS), one log line per seed
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/${1:-r06soak}/soak; mkdir -p $out
for OFF in ${SOAK_SEEDS:-1717 1818}; do
  CAH_TEST_SEED_OFFSET=$OFF timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_multi2.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_small.py tests/test_gpu_long.py -q -m gpu --timeout 600 2>&1 | tail -n 12 > $out/seed_$OFF.log
  echo "seed offset $OFF: $(tail -n 1 $out/seed_$OFF.log)"
done
