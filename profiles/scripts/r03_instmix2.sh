#!/bin/bash
# instruction counts + busy/wait of the prefilter for the libraries named (suffixes as r03_ab.sh)
cd /tmp; export TMPDIR=/tmp; out="$GRAFT_REPO_ROOT/gpurun_out/mix2"; mkdir -p $out
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/cutadapt_amd/libcutadapt_hip${v:+_$v}.so
  [ "$v" = "product" ] && lib=$GRAFT_REPO_ROOT/cutadapt_amd/libcutadapt_hip.so
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_DATA_FIFO_FULL"; do
  rm -rf $out/m
  CAH_LIB_PATH=$lib timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/m -o p -- \
     python "$GRAFT_REPO_ROOT/bench.py" --config C2 --no-other-configs --no-cpu-baseline --check-reads 0 --steps 1 --warmup 0 > $out/m.json 2> $out/m.err
  python - $out/m $v <<'PY'
import csv,glob,sys,collections
path=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(float); dur=0
for r in csv.DictReader(open(path)):
    if "k_filter_stream2" in r["Kernel_Name"]:
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
wr=100e6/64
print(sys.argv[2], "ms", round(dur,3), {k:round(v/wr,1) for k,v in agg.items()})
PY
  done
done
