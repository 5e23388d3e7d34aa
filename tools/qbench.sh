#!/bin/bash
# usage (on the GPU box): tools/qbench.sh [ENV=VAL ...]  -> ms/step and per-stage us of a short bench run
for kv in "$@"; do export "$kv"; done
timeout -k 10 120 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aspp | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$*', 'ms/step %.4f' % d['ms_per_step'], 'Gpix/s %.2f' % d['value'], {k: round(v,1) for k,v in d['roofline']['stage_us'].items()})"
