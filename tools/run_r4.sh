#!/bin/bash
# run on the GPU box: stages of round 4's measurement calls.  tools/run_r4.sh <stage> ...
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for st in "$@"; do
case $st in
  tests)   timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 -p no:cacheprovider > $O/r4_tests.log 2>&1; tail -40 $O/r4_tests.log ;;
  bench)   timeout 900 python bench.py --steps 5 --warmup 2 2>$O/r4_bench.err | tail -1 > $O/r4_bench_n1.json; cut -c1-600 $O/r4_bench_n1.json ;;
  benchdrv) timeout 1200 python bench.py --steps 20 --warmup 5 2>$O/r4_benchdrv.err | tail -1 > $O/r4_bench_driver_cfg_n1.json; cut -c1-400 $O/r4_bench_driver_cfg_n1.json ;;
  idle)    timeout 300 python tools/idle_ramp.py > $O/r4_idle_ramp.txt 2>&1; cat $O/r4_idle_ramp.txt ;;
  dp1)     GENNBV_FORCE_DP=1 GENNBV_FORCE_SHARD=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-flat-rows 2>$O/r4_dp1.err | tail -1 > $O/r4_bench_dp1_n1.json; cut -c1-500 $O/r4_bench_dp1_n1.json ;;
  refdef)  timeout 1200 python bench.py --steps 3 --warmup 1 --height 400 --width 400 --grid 20 --no-flat-rows 2>$O/r4_refdef.err | tail -1 > $O/r4_bench_refdefault_n1.json; cut -c1-600 $O/r4_bench_refdefault_n1.json ;;
  abtrain) timeout 600 python tools/ab_interleaved.py --what train --variant base --variant "base2" --variant "twokernel:GENNBV_FUSED_TRAIN=0" --rounds 12 --json $O/r4_ab_train.json 2>&1 | tail -5 ;;
  abvoxel) timeout 600 python tools/ab_interleaved.py --what voxel --variant base --variant base2 --rounds 20 --json $O/r4_ab_voxel.json 2>&1 | tail -4 ;;
  prof)    bash tools/collect_profiles.sh r04 2>&1 | tail -5 ;;
  mb)      bash tools/prof_minibatch.sh r4_mb 2>&1 | tail -3 ;;
  voxtests) timeout 900 python -m pytest tests/test_voxel_gpu.py tests/test_envstep_gpu.py tests/test_rollout_gpu.py -m gpu -q -x -p no:cacheprovider > $O/r4_voxtests.log 2>&1; tail -15 $O/r4_voxtests.log ;;
  fulltests) timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 ;;
  partests) timeout 900 python -m pytest tests/test_parallel_gpu.py tests/test_ppo_gpu.py -m gpu -q -p no:cacheprovider --durations=5 > $O/r4_partests.log 2>&1; tail -30 $O/r4_partests.log ;;
  abvox2)  timeout 600 python tools/ab_interleaved.py --what voxel --variant "two:GENNBV_VOXEL_FUSED_WALK=0" --variant "fused" --variant "two2:GENNBV_VOXEL_FUSED_WALK=0" --rounds 20 --json $O/r4_ab_voxel.json 2>&1 | tail -6 ;;
  abtrain1) timeout 600 python tools/ab_interleaved.py --what train --variant base --variant base2 --rounds 12 --json $O/r4_ab_train1.json 2>&1 | tail -12 ;;
  abtrain2) timeout 600 python tools/ab_interleaved.py --what train --variant base --variant "twokernel:GENNBV_FUSED_TRAIN=0" --rounds 12 --json $O/r4_ab_train2.json 2>&1 | tail -12 ;;
  voxprof) python tools/microbench_voxel.py 2>&1 | tail -2; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_v; rocprofv3 --kernel-trace --stats -d /tmp/prof_v -- python $GRAFT_REPO_ROOT/tools/microbench_voxel.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_v | grep -E "^kernel|k_|rocclr" | cut -c1-190 | tee $O/r4_voxel_kernel_trace.txt; cd $GRAFT_REPO_ROOT ;;
  dp1prof) GENNBV_FORCE_DP=1 GENNBV_FORCE_SHARD=1 bash tools/prof_minibatch.sh r4_dp1_mb 2>&1 | tail -3 ;;
  enctests) timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_ppo_g64_gpu.py tests/test_range_guard_gpu.py -m gpu -q -x -p no:cacheprovider --durations=8 > $O/r4_enctests.log 2>&1; tail -14 $O/r4_enctests.log ;;
  partests2) timeout 900 python -m pytest tests/test_parallel_gpu.py -m gpu -q -p no:cacheprovider > $O/r4_partests2.log 2>&1; tail -8 $O/r4_partests2.log | cut -c1-300 ;;
  abold)   timeout 900 python tools/ab_interleaved.py --what train --variant "old:LIB=gennbv_amd/libgennbv_hip_old.so" --variant new --variant "old2:LIB=gennbv_amd/libgennbv_hip_old.so" --rounds 16 --json $O/r4_ab_train_old_new.json 2>&1 | tail -8 ;;
  abvoxold) timeout 600 python tools/ab_interleaved.py --what voxel --variant "old:LIB=gennbv_amd/libgennbv_hip_old.so" --variant new --rounds 20 --json $O/r4_ab_voxel_old_new.json 2>&1 | tail -5 ;;
  convprof) cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c; rocprofv3 --kernel-trace --stats -d /tmp/prof_c -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_c | grep -E "^kernel|k_" | cut -c1-190 | tee $O/r4_conv_kernel_trace.txt; cd $GRAFT_REPO_ROOT ;;
  abold3)  timeout 1200 python tools/ab_interleaved.py --what train --captures 3 --variant "old:LIB=gennbv_amd/libgennbv_hip_old.so" --variant new --rounds 10 --json $O/r4_ab_train_old_new_c3.json 2>&1 | grep -v "^\[ab\]" | tail -14 ;;
  partests3) timeout 600 python -m pytest tests/test_parallel_gpu.py -m gpu -q -p no:cacheprovider -k "2-" > $O/r4_partests3.log 2>&1; tail -8 $O/r4_partests3.log | cut -c1-300 ;;
  semantic) timeout 900 python bench.py --steps 3 --warmup 1 --semantic --no-cpu-baseline --no-flat-rows 2>/dev/null | tail -1 > $O/r4_bench_semantic_n1.json; cut -c1-300 $O/r4_bench_semantic_n1.json ;;
  config5) timeout 1500 python bench.py --steps 1 --warmup 1 --envs 512 --grid 128 --no-cpu-baseline --no-flat-rows 2>/dev/null | tail -1 > $O/r4_bench_config5_shard_n1.json; cut -c1-300 $O/r4_bench_config5_shard_n1.json ;;
  rolltests) timeout 900 python -m pytest tests/test_rollout_gpu.py tests/test_fullsize_gpu.py tests/test_ppo_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 ;;
  abroll)  timeout 600 python tools/ab_interleaved.py --what rollout --n-steps 32 --variant graph --variant "eager:ROLLOUT_GRAPH=0" --rounds 8 --json $O/r4_ab_rollout.json 2>&1 | tail -4 ;;
  abvars)  timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant "old:LIB=gennbv_amd/libgennbv_hip_old.so" --variant "adam:LIB=gennbv_amd/libgennbv_hip_adam.so" --variant "adam_nt:LIB=gennbv_amd/libgennbv_hip_adam_nt.so" --rounds 8 --json $O/r4_ab_train_vars.json 2>&1 | grep -v "^\[ab\]" | tail -16 ;;
  ppotests) timeout 900 python -m pytest tests/test_ppo_gpu.py tests/test_rsl_rl_gpu.py tests/test_encoder_gpu.py -m gpu -q -x -p no:cacheprovider -k "adam or Adam or train or conv1_split or fused_train or rsl" 2>&1 | tail -5 ;;
  abnt)    timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant "b0:LIB=gennbv_amd/libgennbv_hip_b0.so" --variant "v1_ldnt:LIB=gennbv_amd/libgennbv_hip_v1.so" --variant "v2_adamnt:LIB=gennbv_amd/libgennbv_hip_v2.so" --rounds 8 --json $O/r4_ab_train_nt.json 2>&1 | grep -v "^\[ab\]" | tail -16 ;;
  abvoxnt) timeout 600 python tools/ab_interleaved.py --what voxel --variant "b0:LIB=gennbv_amd/libgennbv_hip_b0.so" --variant "v3_nt:LIB=gennbv_amd/libgennbv_hip_v3.so" --variant "b0b:LIB=gennbv_amd/libgennbv_hip_b0.so" --rounds 20 --json $O/r4_ab_voxel_nt.json 2>&1 | tail -5 ;;
  abv4)    timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant "b1:LIB=gennbv_amd/libgennbv_hip_b1.so" --variant "v4_i8nt:LIB=gennbv_amd/libgennbv_hip_v4.so" --rounds 8 --json $O/r4_ab_train_v4.json 2>&1 | grep -v "^\[ab\]" | tail -12 ;;
  abrollv4) timeout 900 python tools/ab_interleaved.py --what rollout --n-steps 32 --variant "b1:LIB=gennbv_amd/libgennbv_hip_b1.so" --variant "v4_i8nt:LIB=gennbv_amd/libgennbv_hip_v4.so" --variant "b1b:LIB=gennbv_amd/libgennbv_hip_b1.so" --rounds 8 --json $O/r4_ab_rollout_v4.json 2>&1 | tail -5 ;;
  c5ab)    for L in gennbv_amd/libgennbv_hip_b3.so gennbv_amd/libgennbv_hip.so gennbv_amd/libgennbv_hip_b3.so gennbv_amd/libgennbv_hip.so; do GENNBV_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 900 python bench.py --steps 1 --warmup 1 --envs 512 --grid 128 --no-cpu-baseline --no-flat-rows --no-state-check 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],1), d['config']['breakdown_ms_per_step'])"; done ;;
  g128tests) timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_ppo_g64_gpu.py -m gpu -q -x -p no:cacheprovider -k "128 or fp32" 2>&1 | tail -4 ;;
  abv5)    timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant "b3:LIB=gennbv_amd/libgennbv_hip_b3.so" --variant "vg_gplain:LIB=gennbv_amd/libgennbv_hip_vg.so" --variant "vd_dy2nt:LIB=gennbv_amd/libgennbv_hip_vd.so" --rounds 8 --json $O/r4_ab_train_v5.json 2>&1 | grep -v "^\[ab\]" | tail -14 ;;
  abv6)    timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant "b3:LIB=gennbv_amd/libgennbv_hip_b3.so" --variant "adam16k:LIB=gennbv_amd/libgennbv_hip_ab16384.so" --variant "adam4k:LIB=gennbv_amd/libgennbv_hip_ab4096.so" --rounds 8 --json $O/r4_ab_train_v6.json 2>&1 | grep -v "^\[ab\]" | head -5 ;;
  semprof) cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_s; rocprofv3 --kernel-trace --stats -d /tmp/prof_s -- python $GRAFT_REPO_ROOT/bench.py --semantic --steps 1 --warmup 1 --n-steps 16 --no-cpu-baseline --no-flat-rows --no-state-check > /tmp/prof_s.log 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py /tmp/prof_s 600 k_ppo_fused > $O/r4_semantic_minibatch_timeline.txt; cut -c1-150 $O/r4_semantic_minibatch_timeline.txt; cd $GRAFT_REPO_ROOT ;;
  *) echo "unknown stage $st" ;;
esac
done
