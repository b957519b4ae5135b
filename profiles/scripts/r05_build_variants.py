#!/usr/bin/env python3
"""Developer builds for round 5's A/B runs (CAH_LIB_PATH=...): libcutadapt_hip_<tag>.so next to the product library.
    python profiles/scripts/r05_build_variants.py plain trace ...
Round 4's library for comparison: built from a git worktree of the round's last commit (r05_build_base.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cutadapt_amd import build                                       # noqa: E402

VARIANTS = {
    "plain": ["-DCAH_BS_PLAIN_OPS"],                                  # the compiler's own instruction forms in the scan's column
    "trace": ["-DSCAN_TRACE"],
    "w4": ["-DCAH_SCAN_WAVES=4"],
    "dppplain": ["-DCAH_DPP_PLAIN"],                                  # k_dp_packed with compare + select predicates (round 4's cell)
    "s3w5": ["-DCAH_SCAN3_WAVES=5"],                                  # k_back_scan3 at 5 waves per SIMD (96 VGPRs, 35 spilled)                                     # k_back_scan at 4 waves per SIMD (128 VGPRs)
    "m2w12": ["-DM2_WAVES=12"],                                       # k_multi_stream with 12 waves per CU: 168 VGPRs, no scratch
    "m2w8": ["-DM2_WAVES=8"],                                        # the product's scan with s_memtime stamps
}
for tag in sys.argv[1:]:
    flags = VARIANTS[tag]
    out = os.path.join(os.path.dirname(build.LIB_PATH), f"libcutadapt_hip_{tag}.so")
    print(tag, flags, "->", build.build_library(extra_flags=flags, out_path=out))
