#!/bin/bash
# run on the GPU box: kernel trace of the REFERENCE'S DEFAULT workload (400x400 depth, 20^3 grid) -> timeline of one PPO minibatch + summary
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-refdef}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r
rocprofv3 --kernel-trace --stats -d /tmp/prof_r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --n-steps 16 --height 400 --width 400 --grid 20 --no-cpu-baseline --no-flat-rows --no-state-check > /tmp/prof_r.log 2>&1
echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --n-steps 16 --height 400 --width 400 --grid 20 --no-cpu-baseline --no-flat-rows --no-state-check" > $OUT/refdefault_kernel_trace.txt
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_r >> $OUT/refdefault_kernel_trace.txt
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_r 100 "void k_ppo_fused" > $OUT/refdefault_minibatch_timeline.txt
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_r 5 "void k_hit_list" > $OUT/refdefault_rollout_step_timeline.txt
tail -1 /tmp/prof_r.log | cut -c1-400
