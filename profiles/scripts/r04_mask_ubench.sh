#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04mask
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/scripts/r04_mask_ubench.hip -o /tmp/mask_ubench 2> gpurun_out/r04mask/build.err && /tmp/mask_ubench | tee gpurun_out/r04mask/mask_ubench.txt
