#!/bin/bash
# usage (GPU box): tools/aspp_sweep.sh -> forward / backward-data times of the 11 ASPP convolutions per forced split count
for sp in 0 1 2 3 4 6 9 12 18; do
  echo "== SFM_ASPP_SP=$sp"
  SFM_ASPP_SP=$sp timeout -k 10 200 python3 tools/bench_aspp.py 2>&1 | head -12 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12}'
done
