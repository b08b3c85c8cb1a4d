#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_encoder_gpu.py tests/test_ppo_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/microbench_train.py --backend hip $ENC_ARGS 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_e
rocprofv3 --kernel-trace --stats -d /tmp/prof_e -- python $GRAFT_REPO_ROOT/tools/microbench_train.py --backend hip $ENC_ARGS > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_e | grep -E "^kernel|k_conv|k_bn|k_reduce" | cut -c1-150
