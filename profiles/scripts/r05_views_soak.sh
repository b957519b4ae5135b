#!/bin/bash
# shifted-seed runs of tests/test_gpu_stream.py (the RV form's fuzz of starts and lengths draws from CAH_TEST_SEED_OFFSET)
out=gpurun_out/r05views/soak; mkdir -p $out
for OFF in 707 808 909 1010 1111; do
  CAH_TEST_SEED_OFFSET=$OFF timeout 300 python -m pytest tests/test_gpu_stream.py -q -m gpu 2>&1 | tail -n 3 > $out/views_seed_$OFF.log
  echo "seed offset $OFF: $(tail -n 1 $out/views_seed_$OFF.log)"
done
