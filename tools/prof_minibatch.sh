#!/bin/bash
# run on the GPU box: kernel trace of a short bench run -> timeline of one PPO minibatch + per-kernel summary
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-mb}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --n-steps 16 --no-cpu-baseline --no-flat-rows --no-state-check > /tmp/prof_k.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_k > $OUT/bench_kernel_trace.txt
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_k 600 "void k_ppo_fused" > $OUT/minibatch_timeline.txt
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_k 5 "void k_hit_list" > $OUT/rollout_step_timeline.txt
tail -1 /tmp/prof_k.log | cut -c1-300
