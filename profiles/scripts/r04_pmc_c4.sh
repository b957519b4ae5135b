#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash profiles/scripts/pmc.sh "$1" "" --config C4 --no-other-configs > /dev/null 2>&1
python profiles/summarize_r04.py gpurun_out/$1 $1 > /dev/null 2>&1
cp profiles/r04/$1_* gpurun_out/ 2>/dev/null
python - "$1" <<'PY'
import json,sys
d=json.load(open(f"profiles/r04/{sys.argv[1]}_pmc_summary.json"))
for k,v in d["kernels"].items():
    print(k, {x: (round(v[x],3) if isinstance(v[x],float) else v[x]) for x in ("duration_ms_sq1","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM","SQ_INSTS_BRANCH","SQ_WAIT_ANY_frac_of_wave_cycles","SQ_WAIT_INST_ANY_frac_of_wave_cycles","SQ_ACTIVE_INST_ANY_frac_of_wave_cycles","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_BUSY_CYCLES","valu_busy","waves_per_simd_time_average","hbm_fetch_bytes_x2","hbm_write_bytes","resources") if x in v})
PY
