#!/bin/bash
# usage (on the GPU box): SFM_CASES=1 tools/abl_sizes.sh  -> tools/bench_sizes.py for every variants/lib_*.so, then the base library
cp sfm_amd/libsfm_amd.so /tmp/lib_base.so
for v in variants/lib_*.so; do
  cp $v sfm_amd/libsfm_amd.so
  echo "== $(basename $v .so)"; timeout -k 10 300 python3 -u tools/bench_sizes.py 2>&1 | grep -v amdgpu.ids || exit 1
done
cp /tmp/lib_base.so sfm_amd/libsfm_amd.so
echo "== base"; timeout -k 10 300 python3 -u tools/bench_sizes.py 2>&1 | grep -v amdgpu.ids
