#!/bin/bash
# kernel stats of one config: r03_stats.sh C3 -> gpurun_out/stats_C3/
cfg=${1:-C3}; out="$GRAFT_REPO_ROOT/gpurun_out/stats_$cfg"; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python "$GRAFT_REPO_ROOT/bench.py" --config $cfg --no-other-configs --no-cpu-baseline --check-reads 0 --steps 3 --warmup 1 > $out/trace.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
