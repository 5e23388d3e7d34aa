#!/bin/bash
# VGPR / SGPR / scratch / LDS per kernel of one source file (device assembly, no GPU needed): tools/regs.sh loss.hip off
src=$1; contract=${2:-off}; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize --cuda-device-only -S -ffp-contract=$contract "$@" \
  sfm_amd/csrc/$src -o /tmp/regs_$$.s 2>/dev/null
python3 - /tmp/regs_$$.s <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size", t, re.S):
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    print(f"{g('vgpr_count'):>4} vgpr {g('sgpr_count'):>4} sgpr {g('private_segment_fixed_size'):>5} scratch {g('group_segment_fixed_size'):>6} lds  {g('name')[:110]}")
PY
rm -f /tmp/regs_$$.s
