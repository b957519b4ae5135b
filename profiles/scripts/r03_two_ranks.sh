#!/bin/bash
# the N > 1 launch form the driver uses, on ONE device (ranks share it: --oversubscribe, labelled "not a measurement"):
# shows the multi-rank code path of the round's final code end to end (rendezvous, sharded workload, barrier timing,
# max over ranks, ONE JSON line on rank 0's stdout)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 --oversubscribe --no-other-configs --cpu-seconds 3 > gpurun_out/r3/two_ranks.json 2> gpurun_out/r3/two_ranks.err
echo "rc=$? lines on stdout: $(wc -l < gpurun_out/r3/two_ranks.json)"
head -c 1500 gpurun_out/r3/two_ranks.json; echo; tail -n 5 gpurun_out/r3/two_ranks.err
