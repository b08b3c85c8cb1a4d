#!/bin/bash
# Link a libgennbv_hip.so whose encoder.hip (with conv_split.h) is compiled with extra flags (measurement builds: -DSPLIT_HI_ONLY ...);
# the other objects are the in-tree ones (run `python -m gennbv_amd.csrc.build` first).
#   tools/build_encoder_variant.sh <out.so> <flags...>
set -e
out=$(readlink -f $1); shift
root=$(cd $(dirname $0)/.. && pwd); c=$root/gennbv_amd/csrc
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function "$@" -c $c/encoder.hip -o $tmp/encoder.o
objs=$(ls $c/*.o | grep -v encoder.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $tmp/encoder.o $objs
rm -rf $tmp; echo built $out "$@"
