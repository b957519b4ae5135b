#!/bin/bash
# round 6, one call: the -m gpu suite, smoke(), the seeded parity suites with shifted seeds, the PMC passes of C2..C5 at their
# BASELINE sizes (-> profiles/pmc_latest.json, tied to the library by its build id; per-step sums of every kernel's HBM bytes),
# the default bench line that reads them.  Everything a step writes under profiles/ on the box is copied to gpurun_out/<tag>/.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=${1:-r06final}
out=gpurun_out/$tag; mkdir -p $out/soak
timeout 1800 python -m pytest tests -x -q -m gpu --timeout 900 > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
tail -n 4 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $out/smoke.log
for OFF in ${SOAK_SEEDS:-909 1010}; do
  CAH_TEST_SEED_OFFSET=$OFF timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_multi2.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_small.py tests/test_gpu_long.py -q -m gpu --timeout 600 2>&1 | tail -n 3 > $out/soak/seed_$OFF.log
  echo "seed offset $OFF: $(tail -n 1 $out/soak/seed_$OFF.log)"
done
pmc() {  # config reads tag
  o="$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$1"; mkdir -p "$o"
  args=(--config $1 --no-other-configs)
  ( cd /tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$o/trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" "${args[@]}" --no-cpu-baseline --check-reads 0 --steps 4 --warmup 1 > "$o/trace.json" 2> "$o/trace.err"
    run() { name="$1"; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$o/$name" -o p -- python "$GRAFT_REPO_ROOT/bench.py" "${args[@]}" --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > "$o/$name.json" 2> "$o/$name.err"; }
    run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
    run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
    run sq3 SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
    run fetch FETCH_SIZE
    run write WRITE_SIZE
    run grbm GRBM_GUI_ACTIVE GRBM_COUNT )
  python profiles/summarize_r06.py gpurun_out/${tag}_pmc_$1 $3 --update-latest --config $1 --reads $2 --steps 2 > /dev/null 2> $out/summarize_$1.err
  cp profiles/r06/$3_* $out/ 2>/dev/null
  rm -rf "$o"/*/                                              # (the raw counter CSVs are large: only the summaries travel)
}
pmc C4 100000000 final_c4
pmc C2 100000000 final_c2
pmc C3 100000000 final_c3
pmc C5 125000000 final_c5
cp profiles/pmc_latest.json $out/
t0=$(date +%s); timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python - "$out" <<'PY'
import json,sys
out=sys.argv[1]
try:
    j=json.loads(open(f"{out}/bench_default.json").read().strip().splitlines()[-1])
    print("C2", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, "frac", round(j["roofline"]["frac"],3), j["roofline"]["profile"], "traffic", j["roofline"]["traffic"], "whole", (j["roofline"].get("whole_step_traffic") or {}).get("over_algorithmic"), "cpu", j.get("cpu_baseline",{}).get("value"))
    print(j["config"]["parity_check"][:160])
    for c,o in j.get("other_configs",{}).items():
        print(c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:40], o["roofline"]["kernel"], round(o["roofline"]["frac"],4), o["roofline"].get("traffic"), (o["roofline"].get("whole_step_traffic") or {}).get("over_algorithmic"), (o.get("cpu_baseline") or {}).get("value")))
    for c,o in j.get("p_adapter_extremes",{}).items():
        print(c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:20]))
    for key in ("ragged", "ragged_packed"):
        for c,o in j.get(key,{}).items():
            print(key, c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:20], "x uniform", round(o["vs_uniform"],3), {k:round(v,2) for k,v in o["kernel_ms_per_step"].items()}))
except Exception as e:
    print("FAILED", e); print(open(f"{out}/bench_default.err").read()[-2000:])
PY
