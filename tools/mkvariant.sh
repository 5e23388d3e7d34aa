#!/bin/bash
# usage (CPU side): tools/mkvariant.sh <tag> "<extra hipcc flags>"  -> variants/lib_<tag>.so (travels with gpurun, git-ignored)
set -e
tag=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
cp $root/sfm_amd/csrc/*.hip $root/sfm_amd/csrc/*.h $root/sfm_amd/csrc/Makefile $tmp/
mkdir -p $root/variants
make -C $tmp -j8 EXTRA="$*" TARGET=$root/variants/lib_$tag.so > $tmp/log 2>&1 || { tail -20 $tmp/log; exit 1; }
rm -rf $tmp
echo built variants/lib_$tag.so
