#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib"; rm -rf /tmp/prof_ab
  GENNBV_HIP_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -- python $GRAFT_REPO_ROOT/tools/microbench_train.py --backend hip --iters 5 > /tmp/ab.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_ab | grep -E "k_conv" | cut -c1-130
done
