#!/bin/bash
# Build libgennbv_hip.so of another git revision next to the in-tree one (for tools/ab_interleaved.py --variant "old:LIB=...").
#   tools/build_variant.sh <git-rev> <out.so>
set -e
rev=$1; out=$(readlink -f $2); root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C $root archive $rev gennbv_amd/csrc include | tar -x -C $tmp
cd $tmp/gennbv_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function -c $f -o ${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out *.o
rm -rf $tmp
echo built $out from $rev
