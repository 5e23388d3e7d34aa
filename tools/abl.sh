#!/bin/bash
# usage (on the GPU box): tools/abl.sh  -> qbench of every variants/lib_*.so (timing experiments; the box copy is scratch)
cp sfm_amd/libsfm_amd.so /tmp/lib_base.so
for v in variants/lib_*.so; do
  cp $v sfm_amd/libsfm_amd.so
  tools/qbench.sh TAG=$(basename $v .so) || exit 1
done
cp /tmp/lib_base.so sfm_amd/libsfm_amd.so
tools/qbench.sh TAG=base
