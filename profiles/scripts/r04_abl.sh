#!/bin/bash
# instruction counts of k_multi_stream with parts switched off (developer builds -DM2_ABL=bits: 1 no sweeps, 2 no
# resolve rounds, 4 no events from the main pass): results are wrong, the counters tell where the instructions are
cd /tmp; export TMPDIR=/tmp
for v in "" abl1 abl2 abl3; do
  lib=$GRAFT_REPO_ROOT/cutadapt_amd/libcutadapt_hip${v:+_$v}.so
  out=$GRAFT_REPO_ROOT/gpurun_out/abl/${v:-full}; mkdir -p $out
  CAH_LIB_PATH=$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/bench.py --config C4 --reads 10000000 --no-other-configs --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > $out.json 2> $out.err
  python - "$out" "${v:-full}" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_multi_stream" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg["ms"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
pieces=10_000_000/64
print(sys.argv[2], "ms %.3f"%(sum(agg["ms"])/len(agg["ms"])), {k: round(sum(v)/len(v)/pieces) for k,v in agg.items() if k!="ms"})
PY
done
