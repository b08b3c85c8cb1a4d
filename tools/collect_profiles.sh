#!/bin/bash
# Run on the GPU box: bench line + rocprofv3 kernel-trace summary + HBM traffic counters of the
# voxel kernels.  Writes small text files under gpurun_out/profiles_rNN/ (copy to profiles/).
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_n1.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --n-steps 16 --no-cpu-baseline --no-flat-rows --no-state-check > /tmp/prof_k.log 2>&1
echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --n-steps 16 --no-cpu-baseline --no-flat-rows --no-state-check" > $OUT/bench_kernel_trace.txt
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_k >> $OUT/bench_kernel_trace.txt
# one PPO minibatch / one rollout step as timelines (same trace)
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_k 600 "void k_ppo_fused" > $OUT/minibatch_timeline.txt
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_k 5 "void k_hit_list" > $OUT/rollout_step_timeline.txt
rm -rf /tmp/prof_v
rocprofv3 --kernel-trace --stats -d /tmp/prof_v -- python $GRAFT_REPO_ROOT/tools/microbench_voxel.py > /tmp/prof_v.log 2>&1
echo "# command: rocprofv3 --kernel-trace --stats -- python tools/microbench_voxel.py   (256 envs x 240x320 x 64^3)" > $OUT/voxel_kernel_trace.txt
grep -E "fg frac|update_occ_grid" /tmp/prof_v.log | sed 's/^/# /' >> $OUT/voxel_kernel_trace.txt
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_v | grep -E "^#|^kernel|k_|rocclr" >> $OUT/voxel_kernel_trace.txt
# HBM traffic (separate --pmc passes, no trace domains beside kernel-trace)
: > $OUT/voxel_pmc.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  echo "## --pmc $C" >> $OUT/voxel_pmc.txt
  $GRAFT_REPO_ROOT/tools/run_pmc.sh $OUT/voxel_pmc.txt "k_grid_update|k_hit_list|k_ray_list|k_hit_mask|k_raycast" "$C" -- python $GRAFT_REPO_ROOT/tools/microbench_voxel.py --iters 5
done
python $GRAFT_REPO_ROOT/tools/pmc_to_traffic.py $OUT/voxel_pmc.txt compact > $OUT/voxel_traffic.json
ls -la $OUT
