#!/bin/bash
# instruction / stall counters of the cost scan (k_back_scan3 against k_back_scan), C2 at 50 M reads
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r05pmc; mkdir -p $out
run() {  # tag env name counters...
  tag=$1; envs=$2; name=$3; shift 3
  ( cd /tmp; env $envs timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/${tag}_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --config C2 --reads 50000000 --no-other-configs --no-cpu-baseline --check-reads 0 --steps 2 --warmup 0 > $out/${tag}_$name.json 2> $out/${tag}_$name.err )
}
for v in new old; do
  e="CAH_SCAN3=1"; [ $v = old ] && e="X=1"
  run $v $e sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
  run $v $e sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
done
python - "$out" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for v in ("new","old"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(list)
    for name in ("sq1","sq2"):
        fs=glob.glob(f"{out}/{v}_{name}/**/*counter_collection.csv",recursive=True)
        if not fs: print(v,name,"no csv", open(f"{out}/{v}_{name}.err").read()[-300:]); continue
        seen=set()
        for r in csv.DictReader(open(fs[0])):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if not k.startswith("k_back_scan"): continue
            k=k.split("<")[0]+("<"+r["Kernel_Name"].split("<")[1].split(">")[0]+">" if "<" in r["Kernel_Name"] else "")
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            key=(k,r["Dispatch_Id"])
            if key not in seen and name=="sq1":
                seen.add(key); dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    for k,c in agg.items():
        nl=max(1,len(dur[k]))
        print(v,k,"launches",nl,"ms/launch",round(sum(dur[k])/nl,3), {n:round(x/nl/1e6,2) for n,x in sorted(c.items())})
PY
