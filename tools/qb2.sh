#!/bin/bash
for kv in "$@"; do export "$kv"; done
tools/qbench.sh "$@"
timeout 60 python tools/dbg_fwd.py 2>&1 | grep "flags\|redund"
