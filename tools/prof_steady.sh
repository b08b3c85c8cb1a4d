#!/bin/bash
# run on the GPU box: kernel trace of a LONG bench run (full rollouts, four iterations) -> timeline of a late rollout step and a late
# minibatch: costs that grow with the run (the 100-episode ring, round 6 notes section 12) do not show in the short profile runs
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-steady}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-flat-rows --no-state-check > /tmp/prof_s.log 2>&1
echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-flat-rows --no-state-check   (the 450th env step / 5000th minibatch of the process)" > $OUT/steady_rollout_step_timeline.txt
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_s 450 "void k_hit_list" >> $OUT/steady_rollout_step_timeline.txt
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_s 5000 "void k_ppo_fused" > $OUT/steady_minibatch_timeline.txt
tail -1 /tmp/prof_s.log | cut -c1-300
