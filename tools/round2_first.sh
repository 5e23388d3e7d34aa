#!/bin/bash
# first GPU call of round 2: the WIP no-alloc test, the whole GPU suite, the bench line, a c2-only kernel profile
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/r02_noalloc.log 2>&1; rc=$?
tail -5 gpurun_out/r02_noalloc.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02_gputests.log 2>&1; rc=$?
tail -8 gpurun_out/r02_gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; rc=$?
tail -c 3000 gpurun_out/r02_bench_a.json
[ $rc -eq 0 ] || { tail -20 gpurun_out/r02_bench_a.err; exit $rc; }
SFM_STENCIL_FORK=0 tools/qbench.sh TAG=nofork || exit 1
tools/qbench.sh TAG=fork || exit 1
tools/prof.sh r02a
