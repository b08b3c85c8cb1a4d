#!/bin/bash
# Run on the GPU box: the measurement stages of round 4.   tools/run_r4.sh <stage> [<stage> ...]
# Outputs land under gpurun_out/ (copy what is to be kept into profiles/).  Variant libraries for the A/B stages are built beforehand with
# tools/build_variant.sh <git-rev> gennbv_amd/libgennbv_hip_<name>.so (they travel with the gpurun snapshot; delete them afterwards).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for st in "$@"; do
case $st in
  tests)    timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 -p no:cacheprovider > $O/r4_tests.log 2>&1; tail -40 $O/r4_tests.log ;;
  bench)    timeout 900 python bench.py --steps 5 --warmup 2 2>$O/r4_bench.err | tail -1 > $O/r4_bench_n1.json; cut -c1-600 $O/r4_bench_n1.json ;;
  benchdrv) timeout 1200 python bench.py --steps 20 --warmup 5 2>$O/r4_benchdrv.err | tail -1 > $O/r4_bench_driver_cfg_n1.json; cut -c1-400 $O/r4_bench_driver_cfg_n1.json ;;
  dp1)      GENNBV_FORCE_DP=1 GENNBV_FORCE_SHARD=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-flat-rows 2>$O/r4_dp1.err | tail -1 > $O/r4_bench_dp1_n1.json; cut -c1-500 $O/r4_bench_dp1_n1.json ;;
  refdef)   timeout 1200 python bench.py --steps 3 --warmup 1 --height 400 --width 400 --grid 20 --no-flat-rows 2>$O/r4_refdef.err | tail -1 > $O/r4_bench_refdefault_n1.json; cut -c1-600 $O/r4_bench_refdefault_n1.json ;;
  semantic) timeout 900 python bench.py --steps 3 --warmup 1 --semantic --no-cpu-baseline --no-flat-rows 2>/dev/null | tail -1 > $O/r4_bench_semantic_n1.json; cut -c1-300 $O/r4_bench_semantic_n1.json ;;
  config5)  timeout 1500 python bench.py --steps 1 --warmup 1 --envs 512 --grid 128 --no-cpu-baseline --no-flat-rows 2>/dev/null | tail -1 > $O/r4_bench_config5_shard_n1.json; cut -c1-300 $O/r4_bench_config5_shard_n1.json ;;
  idle)     timeout 300 python tools/idle_ramp.py > $O/r4_idle_ramp.txt 2>&1; cat $O/r4_idle_ramp.txt ;;
  hostprof) timeout 300 python tools/profile_rollout_host.py 2>&1 | tail -60 | cut -c1-170 ;;
  prof)     bash tools/collect_profiles.sh r04 2>&1 | tail -5 ;;   # bench + kernel trace + timelines + voxel trace / PMC -> gpurun_out/profiles_r04/
  convprof) cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c; rocprofv3 --kernel-trace --stats -d /tmp/prof_c -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_c | grep -E "^kernel|k_" | cut -c1-190 | tee $O/r4_conv_kernel_trace.txt; cd $GRAFT_REPO_ROOT ;;
  dp1prof)  GENNBV_FORCE_DP=1 GENNBV_FORCE_SHARD=1 bash tools/prof_minibatch.sh r4_dp1_mb 2>&1 | tail -3 ;;
  semprof)  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_s; rocprofv3 --kernel-trace --stats -d /tmp/prof_s -- python $GRAFT_REPO_ROOT/bench.py --semantic --steps 1 --warmup 1 --n-steps 16 --no-cpu-baseline --no-flat-rows --no-state-check > /tmp/prof_s.log 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_s 600 k_ppo_fused > $O/r4_semantic_minibatch_timeline.txt; cut -c1-150 $O/r4_semantic_minibatch_timeline.txt; cd $GRAFT_REPO_ROOT ;;
  refprof)  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r; rocprofv3 --kernel-trace --stats -d /tmp/prof_r -- python $GRAFT_REPO_ROOT/bench.py --height 400 --width 400 --grid 20 --steps 1 --warmup 1 --n-steps 16 --no-cpu-baseline --no-flat-rows --no-state-check > /tmp/prof_r.log 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_r 600 k_ppo_fused > $O/r4_refdefault_minibatch_timeline.txt; cut -c1-150 $O/r4_refdefault_minibatch_timeline.txt; cd $GRAFT_REPO_ROOT ;;
  # same-process A/Bs (tools/ab_interleaved.py): OLD = gennbv_amd/libgennbv_hip_old.so built from the revision to compare against
  abtrain)  timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant "old:LIB=gennbv_amd/libgennbv_hip_old.so" --variant new --rounds 8 --json $O/r4_ab_train.json 2>&1 | grep -v "^\[ab\]" | tail -12 ;;
  abvoxel)  timeout 600 python tools/ab_interleaved.py --what voxel --variant "old:LIB=gennbv_amd/libgennbv_hip_old.so" --variant new --variant "old2:LIB=gennbv_amd/libgennbv_hip_old.so" --rounds 20 --json $O/r4_ab_voxel.json 2>&1 | tail -5 ;;
  abrollout) timeout 900 python tools/ab_interleaved.py --what rollout --n-steps 32 --variant "old:LIB=gennbv_amd/libgennbv_hip_old.so" --variant new --variant "old2:LIB=gennbv_amd/libgennbv_hip_old.so" --rounds 8 --json $O/r4_ab_rollout.json 2>&1 | tail -5 ;;
  *) echo "unknown stage $st" ;;
esac
done
