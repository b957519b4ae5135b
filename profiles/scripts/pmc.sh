#!/bin/bash
# usage: pmc.sh <outdir under gpurun_out> <env assignments or ''> <bench.py args...>
# One rocprofv3 pass per counter group (PMC passes carry --kernel-trace only; MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass), CSV output; profiles/summarize_r03.py condenses the result.
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
envs="$1"; shift
mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
    name="$1"; shift
    env $envs timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o p -- \
        python "$GRAFT_REPO_ROOT/bench.py" "${BENCH_ARGS[@]}" > "$out/$name.json" 2> "$out/$name.err"
}
BENCH_ARGS=("$@" --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0)
env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- \
    python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline --check-reads 0 --steps 5 --warmup 1 > "$out/trace.json" 2> "$out/trace.err"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq3 SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_LDS_DATA_FIFO_FULL
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd "$GRAFT_REPO_ROOT"
find "$out" -name "*.csv" | head -30
