#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <tag> "<counters>" [script [args...]]  -> per-kernel mean of each counter (one pass = one run)
set -e
tag=$1; ctr=$2; script=${3:-bench.py}
if [ $# -ge 3 ]; then shift 3; else shift $#; fi
args=${@:---steps 10 --warmup 2 --no-cpu-baseline --no-aspp}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/$script $args > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'sfm' not in k and 'loss' not in k: continue
    print(f"{k:48s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
PY
