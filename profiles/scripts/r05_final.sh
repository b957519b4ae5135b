#!/bin/bash
# end of round 5, one call: the -m gpu suite, smoke(), the PMC passes of C2..C5 at their BASELINE sizes (-> profiles/pmc_latest.json,
# tied to the sources by their hash), the default bench line that reads them, then the instruction counts of k_multi_stream
# with parts switched off (developer builds, libcutadapt_hip_abl<bits>.so, if present).  Everything a step writes under
# profiles/ on the box is copied to gpurun_out/r05final/ (only gpurun_out/ travels back).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05final; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
tail -n 4 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.log
# ---- soak: the parity suites that draw their cases from seeded generators, with shifted seeds (conftest.py:
# CAH_TEST_SEED_OFFSET), and the parked k_back_scan3 switched on for every plan that can take it
mkdir -p $out/soak
for OFF in 505 606; do
  CAH_TEST_SEED_OFFSET=$OFF timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_multi2.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_small.py tests/test_gpu_long.py -q -m gpu --timeout 600 2>&1 | tail -n 3 > $out/soak/seed_$OFF.log
  echo "seed offset $OFF: $(tail -n 1 $out/soak/seed_$OFF.log)"
done
CAH_SCAN3=1 timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_configs.py tests/test_gpu_small.py tests/test_gpu_dropin.py -q -m gpu --timeout 600 2>&1 | tail -n 3 > $out/soak/scan3_on.log
echo "CAH_SCAN3=1: $(tail -n 1 $out/soak/scan3_on.log)"
# ---- PMC: trace + sq1 sq2 fetch write grbm (profiles/scripts/pmc.sh without its third SQ pass)
pmc() {  # config reads tag
  o="$GRAFT_REPO_ROOT/gpurun_out/r05final_pmc_$1"; mkdir -p "$o"
  args=(--config $1 --no-other-configs)
  ( cd /tmp
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$o/trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" "${args[@]}" --no-cpu-baseline --check-reads 0 --steps 4 --warmup 1 > "$o/trace.json" 2> "$o/trace.err"
    run() { name="$1"; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$o/$name" -o p -- python "$GRAFT_REPO_ROOT/bench.py" "${args[@]}" --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > "$o/$name.json" 2> "$o/$name.err"; }
    run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
    run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
    run fetch FETCH_SIZE
    run write WRITE_SIZE
    run grbm GRBM_GUI_ACTIVE GRBM_COUNT )
  python profiles/summarize_r05.py gpurun_out/r05final_pmc_$1 $3 --update-latest --config $1 --reads $2 > /dev/null 2> $out/summarize_$1.err
  cp profiles/r05/$3_* $out/ 2>/dev/null
  rm -rf "$o"/*/                                              # (the raw counter CSVs are large: only the summaries travel)
}
pmc C4 100000000 final_c4
pmc C2 100000000 final_c2
pmc C3 100000000 final_c3
pmc C5 125000000 final_c5
cp profiles/pmc_latest.json $out/
# ---- views inside a uniform batch (RV form) against uniform reads and the packed layout, same box
PACKED=1 bash profiles/scripts/r05_views.sh > $out/views_run.txt 2>&1; tail -n 5 $out/views_run.txt; cp gpurun_out/r05views/c2_uniform.json gpurun_out/r05views/c2_views.json gpurun_out/r05views/c2_packed.json gpurun_out/r05views/c5_views.json $out/ 2>/dev/null
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python - "$out" <<'PY'
import json,sys
out=sys.argv[1]
try:
    j=json.loads(open(f"{out}/bench_default.json").read().strip().splitlines()[-1])
    print("C2", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, "frac", round(j["roofline"]["frac"],3), j["roofline"]["profile"], "traffic", j["roofline"]["traffic"], "cpu", j.get("cpu_baseline",{}).get("value"))
    for c,o in j.get("other_configs",{}).items():
        print(c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:40], o["roofline"]["kernel"], round(o["roofline"]["frac"],4), o["roofline"].get("traffic"), (o.get("cpu_baseline") or {}).get("value")))
    for c,o in j.get("p_adapter_extremes",{}).items():
        print(c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:20]))
    for c,o in j.get("ragged",{}).items():
        print("ragged", c, o.get("error") or (round(o["value"]), round(o["ms_per_step"],2), o["parity_check"][:20], "x uniform", round(o["vs_uniform"],3)))
except Exception as e:
    print("FAILED", e); print(open(f"{out}/bench_default.err").read()[-2000:])
PY

