#!/bin/bash
# run on the GPU box: conv-stack microbench (B = 128, G = 64) under "ENV=val;ENV=val" configurations, per-kernel averages
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  echo "== $cfg"
  rm -rf /tmp/prof_c
  env $(echo $cfg | tr ';' ' ') rocprofv3 --kernel-trace --stats -d /tmp/prof_c -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py > /tmp/abc.log 2>&1
  grep "per conv-stack" /tmp/abc.log
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_c | grep -E "k_conv|k_bn|k_reduce|k_c1w|k_stats" | cut -c1-150
done
