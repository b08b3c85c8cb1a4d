#!/bin/bash
# usage: tools/run_pmc.sh <out.txt> <kernel-regex> "<counters...>" -- <cmd...>   (run on the GPU box)
out=$1; shift; rx=$1; shift; ctrs=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_run
rocprofv3 --kernel-trace --kernel-include-regex "$rx" --pmc $ctrs -d /tmp/pmc_run -- "$@" > /tmp/pmc_run.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_run "" >> $out 2>&1
