#!/bin/bash
# usage (on the GPU box): tools/sweep_rows.sh "8 11 16" "12 15"  -> qbench per SFM_STENCIL_ROWS / SFM_STENCIL_ROWS_G value
for r in $1; do tools/qbench.sh SFM_STENCIL_ROWS=$r || exit 1; done
for r in $2; do tools/qbench.sh SFM_STENCIL_ROWS_G=$r || exit 1; done
