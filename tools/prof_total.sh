#!/bin/bash
# usage (on the GPU box): tools/prof_total.sh <tag>  -> per-kernel averages of tools/bench_total.py
tag=${1:-t}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_total_$tag
rm -rf $out
N_STEPS=20 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_total.py > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
tot = 0
for r in list(csv.DictReader(open(sys.argv[1]))):
    per_step = float(r["TotalDurationNs"]) / 30 / 1e3
    tot += per_step
    if per_step > 2: print(f"{r['Name'][:72]:72s} calls/step {int(r['Calls'])/30:5.1f} avg_us {float(r['AverageNs'])/1e3:8.2f} us/step {per_step:8.2f}")
print("sum of kernel time per step (us):", round(tot, 1))
PY
