#!/bin/bash
# round 4's library (and its SCAN_TRACE build) from a worktree of commit c680ea3, as libcutadapt_hip_base.so / _trace_base.so
set -e
cd "$(git rev-parse --show-toplevel)"
wt=/tmp/cah_base_wt
[ -d $wt ] || git worktree add --detach $wt c680ea3 >/dev/null
( cd $wt && python - <<'PY'
from cutadapt_amd import build
import os
here = os.path.dirname(build.LIB_PATH)
print(build.build_library(extra_flags=["-DCAH_R04"], out_path=os.path.join(here, "libcutadapt_hip_base.so")))
print(build.build_library(extra_flags=["-DSCAN_TRACE"], out_path=os.path.join(here, "libcutadapt_hip_trace_base.so")))
PY
)
cp $wt/cutadapt_amd/libcutadapt_hip_base.so $wt/cutadapt_amd/libcutadapt_hip_trace_base.so cutadapt_amd/
