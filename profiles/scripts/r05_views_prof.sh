#!/bin/bash
# one gpurun call: rocprofv3 kernel-trace stats + PMC passes (SQ groups, FETCH_SIZE, WRITE_SIZE, GRBM) of bench.py --config C2
# --ragged (views inside a uniform batch: k_filter_stream2's RV form) -> profiles/r05/views_c2_{kernel_stats.csv,pmc_summary.json}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05views; mkdir -p $out
o="$GRAFT_REPO_ROOT/gpurun_out/r05views_pmc"; mkdir -p "$o"
args=(--config C2 --ragged --no-other-configs)
( cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$o/trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" "${args[@]}" --no-cpu-baseline --check-reads 0 --steps 4 --warmup 1 > "$o/trace.json" 2> "$o/trace.err"
  run() { name="$1"; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$o/$name" -o p -- python "$GRAFT_REPO_ROOT/bench.py" "${args[@]}" --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > "$o/$name.json" 2> "$o/$name.err"; }
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
  run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run grbm GRBM_GUI_ACTIVE GRBM_COUNT )
python profiles/summarize_r05.py gpurun_out/r05views_pmc views_c2 > /dev/null 2> $out/summarize_views.err
cp profiles/r05/views_c2_* $out/ 2>/dev/null
rm -rf "$o"/*/
ls $out | tr '\n' ' '; tail -2 $out/summarize_views.err
