#!/bin/bash
# instruction counts of the C4 kernels, round 5's library against the current one (20 M reads, counters per 64 reads)
cd /tmp; export TMPDIR=/tmp
for v in r05 prod "$@"; do
  lib=$GRAFT_REPO_ROOT/gpurun_in/lib_$v.so; [ "$v" = "prod" ] && lib=$GRAFT_REPO_ROOT/cutadapt_amd/libcutadapt_hip.so
  out=$GRAFT_REPO_ROOT/gpurun_out/r06pmc/$v; mkdir -p $out
  CAH_LIB_ANY_ABI=1 CAH_LIB_PATH=$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/bench.py --config C4 --reads 20000000 --no-other-configs --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > $out.json 2> $out.err
  python - "$out" "$v" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if k.startswith("k_multi") or k.startswith("k_dp"):
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["ms"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
pieces=20_000_000/64
for k,a in agg.items():
    print(sys.argv[2], k, "ms %.3f"%(sum(a["ms"])/len(a["ms"])), {c: round(sum(v)/len(v)/pieces) for c,v in a.items() if c!="ms"})
PY
  rm -rf $out
done
