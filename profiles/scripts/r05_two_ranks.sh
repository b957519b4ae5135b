#!/bin/bash
# the driver's N = 2 launch form on the round's final sources, both ranks on the box's one device (--oversubscribe:
# labelled "not a measurement" in the line): rendezvous, sharded workload, barrier timing, ONE JSON line
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r05last; mkdir -p $out
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 --oversubscribe --no-other-configs --cpu-seconds 3 --reads 40000000 > $out/two_ranks.json 2> $out/two_ranks.err
echo "two ranks rc=$? lines on stdout: $(wc -l < $out/two_ranks.json)"; head -c 400 $out/two_ranks.json; echo
