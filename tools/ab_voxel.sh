#!/bin/bash
# run on the GPU box: time the k_* kernels for several "ENV=val,lib" configurations
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  lib=${cfg##*,}; envs=${cfg%,*}; [ "$envs" == "$cfg" ] && envs=""
  echo "== $cfg"
  rm -rf /tmp/prof_ab
  env $(echo $envs | tr ';' ' ') GENNBV_HIP_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -- python $GRAFT_REPO_ROOT/tools/microbench_voxel.py > /tmp/ab.log 2>&1
  grep "update_occ_grid" /tmp/ab.log
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_ab | grep -E "k_ray|k_hit|k_grid|fillBuffer" | cut -c1-150
done
