#!/bin/bash
# A/B of the conv stack with / without the split-f16 conv2 kernels (per-kernel averages from rocprofv3).
cd /tmp && export TMPDIR=/tmp
for v in ${1:-0 1}; do
  rm -rf /tmp/prof_c$v
  GENNBV_CONV_SPLIT=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_c$v -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py > /tmp/prof_c$v.log 2>&1
  echo "== GENNBV_CONV_SPLIT=$v"; grep "per conv-stack" /tmp/prof_c$v.log
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_c$v | grep -E "k_conv|k_prep|k_bn|k_reduce|k_stats|k_c1w" | cut -c1-200
done
