#!/bin/bash
# round 4, first look at the streaming multi-adapter path: its GPU tests, then C4 bench old vs new in one call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_multi2.py -x -q > gpurun_out/r04a/tests_multi2.log 2>&1; echo "multi2 tests rc $?"; tail -15 gpurun_out/r04a/tests_multi2.log
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_configs.py -x -q > gpurun_out/r04a/tests_multi.log 2>&1; echo "multi/config tests rc $?"; tail -5 gpurun_out/r04a/tests_multi.log
for v in new old; do
  [ $v = old ] && export CAH_NO_MULTI2=1 || unset CAH_NO_MULTI2
  timeout 600 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 200000 > gpurun_out/r04a/bench_c4_$v.json 2> gpurun_out/r04a/bench_c4_$v.err
  echo "bench $v rc $?"; tail -c 1500 gpurun_out/r04a/bench_c4_$v.json; tail -3 gpurun_out/r04a/bench_c4_$v.err
done
