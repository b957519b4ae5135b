#!/bin/bash
# instruction counts of the prefilter with parts of it switched off (CAH_S2_NOMATCH: 1 copy only, 2 copy without
# result rows, 4 everything but the result rows): where do the VALU instructions of a piece go?
cd /tmp; export TMPDIR=/tmp; out="$GRAFT_REPO_ROOT/gpurun_out/mix"; mkdir -p $out
for v in 0 1 2 4; do
  CAH_S2_NOMATCH=$v timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES --kernel-trace --output-format csv -d $out/m$v -o p -- \
     python "$GRAFT_REPO_ROOT/bench.py" --config C2 --no-other-configs --no-cpu-baseline --check-reads 0 --steps 1 --warmup 0 > $out/m$v.json 2> $out/m$v.err
  python - $out/m$v $v <<'PY'
import csv,glob,sys,collections
path=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(float); dur=[]
for r in csv.DictReader(open(path)):
    if "k_filter_stream2" in r["Kernel_Name"]:
        agg[r["Counter_Name"]]+=float(r["Counter_Value"])
wr=100e6/64
print("NOMATCH",sys.argv[2],{k:round(v/wr,1) for k,v in agg.items()})
PY
done
