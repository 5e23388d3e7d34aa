#!/bin/bash
# usage (on the GPU box): tools/abl_aspp.sh  -> bench_aspp conv lines for every variants/lib_*.so (the box copy is scratch)
for v in variants/lib_*.so; do
  cp $v sfm_amd/libsfm_amd.so
  echo "== $(basename $v .so)"
  timeout -k 10 200 python3 tools/bench_aspp.py 2>&1 | grep -E "^[0-9]+x[0-9]+ [0-9]+->" | cut -c1-120 || exit 1
done
