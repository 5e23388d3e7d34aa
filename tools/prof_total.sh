#!/bin/bash
# usage (on the GPU box): tools/prof_total.sh <tag>  -> per-kernel averages of the total-loss leg (pyramid kernels included)
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/proft_$tag
rm -rf $out
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_total.py > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f}  {float(r['Percentage']):6.2f}%")
PY
tail -12 $out.log
