#!/bin/bash
# run on the GPU box: parity tests for the voxel path, microbench, kernel-trace summary (k_* rows only)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_voxel_gpu.py -m gpu -q -x 2>&1 | tail -4
python tools/microbench_voxel.py "$@" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_v
rocprofv3 --kernel-trace --stats -d /tmp/prof_v -- python $GRAFT_REPO_ROOT/tools/microbench_voxel.py "$@" > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_v | grep -E "^kernel|k_|rocclr" | cut -c1-190
