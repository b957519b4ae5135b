#!/bin/bash
# the -m gpu suite, smoke(), and the seeded parity suites with shifted seeds (conftest.py: CAH_TEST_SEED_OFFSET)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06suite; mkdir -p $out/soak
timeout 1800 python -m pytest tests -x -q -m gpu --timeout 900 > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
tail -n 6 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $out/smoke.log
for OFF in ${SOAK_SEEDS:-707 808}; do
  CAH_TEST_SEED_OFFSET=$OFF timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_multi2.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_small.py tests/test_gpu_long.py -q -m gpu --timeout 600 2>&1 | tail -n 3 > $out/soak/seed_$OFF.log
  echo "seed offset $OFF: $(tail -n 1 $out/soak/seed_$OFF.log)"
done
