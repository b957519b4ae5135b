#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_gpu_multi2.py -x -q 2>&1 | tail -3
bash profiles/scripts/r04_c4_quick.sh
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/abl/now; mkdir -p $out
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/bench.py --config C4 --reads 10000000 --no-other-configs --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > $out.json 2> $out.err
python - "$out" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "k_multi" in k or "k_dp" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["ms"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
pieces=10_000_000/64
for k,d in agg.items():
    print(k, "ms %.3f"%(sum(d["ms"])/len(d["ms"])), {c: round(sum(v)/len(v)/pieces) for c,v in d.items() if c!="ms"})
PY
