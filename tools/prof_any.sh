#!/bin/bash
# usage (on the GPU box): tools/prof_any.sh <tag> <script.py> [args...]  -> per-kernel totals of one python script
set -e
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/$script "$@" > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):6.2f}%")
PY
tail -2 $out.log
