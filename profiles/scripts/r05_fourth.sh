#!/bin/bash
# round 5, fourth call: the parked k_back_scan3's GPU test, C4 with k_multi_stream at 12 / 8 waves per CU (no scratch),
# the N = 2 launch form on one device, and the clocks under load
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r05e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_multi2.py -x -q -m gpu --timeout 600 > $out/tests.log 2>&1; tail -n 6 $out/tests.log | cut -c1-300
ab() {  # tag lib config steps
  CAH_LIB_PATH=$2 timeout 400 python bench.py --config $3 --steps $4 --warmup 1 --no-cpu-baseline --no-other-configs --check-reads 200000 > $out/b_$1.json 2> $out/b_$1.err
  python - "$1" "$out/b_$1" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]+".json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(j["value"]), round(j["ms_per_step"],3), {k:round(v,3) for k,v in j["roofline"]["kernel_ms_per_step"].items()}, j["config"]["parity_check"][:30])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]+".err").read()[-800:])
PY
}
P=$PWD/cutadapt_amd
for rep in 1 2; do
  ab c4_w16_$rep $P/libcutadapt_hip.so C4 3
  ab c4_w12_$rep $P/libcutadapt_hip_m2w12.so C4 3
  ab c4_w8_$rep $P/libcutadapt_hip_m2w8.so C4 3
done
# ---- the driver's N = 2 launch form, oversubscribed on the one device (not a measurement: the code path end to end)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 --oversubscribe --no-other-configs --cpu-seconds 3 --reads 40000000 > $out/two_ranks.json 2> $out/two_ranks.err
echo "two ranks rc=$? lines on stdout: $(wc -l < $out/two_ranks.json)"; head -c 600 $out/two_ranks.json; echo
# ---- clocks and power while the headline kernels run (200 steps of C2), sampled every 100 ms
( for i in $(seq 1 120); do /opt/rocm/bin/rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > $out/clocks.jsonl &
SMI=$!
timeout 300 python bench.py --steps 200 --warmup 3 --no-cpu-baseline --no-other-configs --check-reads 0 > $out/b_c2_200.json 2> $out/b_c2_200.err
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python - "$out" <<'PY'
import json,sys
out=sys.argv[1]
try:
    j=json.loads(open(out+"/b_c2_200.json").read().strip().splitlines()[-1]); print("C2 x 200 steps:", round(j["value"]), round(j["ms_per_step"],3))
except Exception as e: print("bench failed", e)
rows=[]
for line in open(out+"/clocks.jsonl"):
    try: d=json.loads(line)
    except Exception: continue
    for card,v in d.items():
        if not isinstance(v, dict): continue
        rows.append({k:v[k] for k in v if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
print(len(rows), "samples; first/last/middle:")
for r in rows[:2]+rows[len(rows)//2:len(rows)//2+3]+rows[-2:]: print("  ", r)
PY
