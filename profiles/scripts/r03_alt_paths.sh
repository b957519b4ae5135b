#!/bin/bash
# the whole -m gpu suite with the alternative code paths forced (none of these switches may change a result)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/alt; export TMPDIR=/tmp
for v in "CAH_NO_UNIFORM=1" "CAH_NO_STREAM2=1" "CAH_NO_LINKED_FUSE=1" "CAH_NO_FILTER_CLEAR=1" "CAH_S2_GLOBAL=1"; do
  env $v timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/alt/$v.log 2>&1
  echo "$v rc=$? $(tail -n 1 gpurun_out/alt/$v.log)"
done
