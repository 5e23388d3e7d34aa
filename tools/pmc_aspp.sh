#!/bin/bash
# usage (on the GPU box): tools/pmc_aspp.sh <tag>  -> MFMA-utilisation counters of the shipped ASPP kernels (one pass each)
# layers: 32x104 256->128 r2 (128x128 tiles) and 128x416 64->32 r3 (256x32 tiles), forward and weight gradient
tag=${1:-r}
out=gpurun_out/pmc_aspp_$tag.txt
: > $out
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_BUSY_CU_CYCLES|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_INSTS_VALU|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY" | sort -u > gpurun_out/pmc_aspp_available_$tag.txt
for layer in "32 104 256 128 2" "128 416 64 32 3"; do
  for what in fwd bwd_weight; do
    for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
      echo "== layer $layer $what : $ctr" >> $out
      tools/pmc.sh aspp_$tag "$ctr" tools/one_conv.py $layer $what >> $out 2>&1 || { echo "pass failed" >> $out; tail -3 gpurun_out/pmc_aspp_$tag.log >> $out; }
    done
  done
done
cat $out
