#!/bin/bash
# Run on the GPU box: the measurement stages of round 5.   tools/run_r5.sh <stage> [<stage> ...]   (outputs under gpurun_out/)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
OLD=gennbv_amd/libgennbv_hip_old.so
for st in "$@"; do
case $st in
  tests)     timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 -p no:cacheprovider > $O/r5_tests.log 2>&1; tail -30 $O/r5_tests.log ;;
  tests_sel) timeout 1200 python -m pytest tests/test_rollout_gpu.py tests/test_fullsize_gpu.py tests/test_ppo_gpu.py tests/test_ppo_g64_gpu.py tests/test_encoder_gpu.py tests/test_rsl_rl_gpu.py -m gpu -q --maxfail=6 --durations=6 -p no:cacheprovider > $O/r5_tests_sel.log 2>&1; tail -25 $O/r5_tests_sel.log ;;
  bench)     timeout 900 python bench.py --steps 5 --warmup 2 2>$O/r5_bench.err | tail -1 > $O/r5_bench_n1.json; cut -c1-600 $O/r5_bench_n1.json ;;
  benchdrv)  timeout 1200 python bench.py --steps 20 --warmup 5 2>$O/r5_benchdrv.err | tail -1 > $O/r5_bench_driver_cfg_n1.json; cut -c1-400 $O/r5_bench_driver_cfg_n1.json ;;
  dpab)      timeout 1200 python -m pytest tests/test_parallel_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
             for v in ${DPAB_VARIANTS:-"new:" "old:GENNBV_DP_LATE_ASIDE=0,GENNBV_DP_ROTATE=0" "new2:"}; do n=${v%%:*}; e=${v#*:}; env ${e//,/ } GENNBV_FORCE_DP=1 GENNBV_FORCE_SHARD=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-flat-rows 2>$O/r5_dp1_$n.err | tail -1 > $O/r5_bench_dp1_$n.json; python - <<PY
import json
d=json.load(open("$O/r5_bench_dp1_$n.json")); print("$n", round(d["ms_per_step"],1), "ms", d["train_roofline"]["ms_per_minibatch"], d.get("timed_state_check"), d["config"].get("dp_graph_mode"))
PY
             done ;;
  dp1)       GENNBV_FORCE_DP=1 GENNBV_FORCE_SHARD=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-flat-rows 2>$O/r5_dp1.err | tail -1 > $O/r5_bench_dp1_n1.json; cut -c1-500 $O/r5_bench_dp1_n1.json ;;
  prof)      bash tools/collect_profiles.sh r05 2>&1 | tail -5 ;;
  refdef)    timeout 1200 python bench.py --steps 3 --warmup 1 --height 400 --width 400 --grid 20 --no-flat-rows 2>$O/r5_refdef.err | tail -1 > $O/r5_bench_refdefault_n1.json; cut -c1-400 $O/r5_bench_refdefault_n1.json ;;
  semantic)  timeout 900 python bench.py --steps 3 --warmup 1 --semantic --no-cpu-baseline --no-flat-rows 2>/dev/null | tail -1 > $O/r5_bench_semantic_n1.json; cut -c1-300 $O/r5_bench_semantic_n1.json ;;
  config5)   timeout 1500 python bench.py --steps 1 --warmup 1 --envs 512 --grid 128 --no-cpu-baseline --no-flat-rows 2>/dev/null | tail -1 > $O/r5_bench_config5_shard_n1.json; cut -c1-300 $O/r5_bench_config5_shard_n1.json ;;
  abvoxrot)  timeout 300 python tools/ab_interleaved.py --what voxel --variant base --variant "rot:LIB=gennbv_amd/libgennbv_hip_rot.so" --variant base2 --rounds 20 --json $O/r05_ab_voxel_chunk_rot.json 2>&1 | tail -4 ;;
  abrollrot) timeout 600 python tools/ab_interleaved.py --what rollout --n-steps 32 --variant base --variant "rot:LIB=gennbv_amd/libgennbv_hip_rot.so" --variant base2 --rounds 6 --json $O/r05_ab_rollout_chunk_rot.json 2>&1 | tail -4 ;;
  abtail)    timeout 1200 python tools/ab_interleaved.py --what train --captures 3 --variant "two:GENNBV_TAIL_MERGE=0" --variant merged --rounds 8 --json $O/r05_ab_train_tail_merge.json 2>&1 | grep -v "^\[ab\]" | tail -12 ;;
  abside)    GENNBV_WGRAD_FINISH_SIDE=1 timeout 900 python -m pytest tests/test_ppo_g64_gpu.py -x -q -m gpu 2>&1 | tail -3; timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant base --variant "side:GENNBV_WGRAD_FINISH_SIDE=1" --rounds 8 --json $O/r05_ab_train_wgrad_finish_second_branch.json 2>&1 | grep -v "^\[ab\]" | tail -12 ;;
  abtrain)   timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant "old:LIB=$OLD" --variant new --rounds 8 --json $O/r05_ab_train.json 2>&1 | grep -v "^\[ab\]" | tail -12 ;;
  abvoxel)   timeout 600 python tools/ab_interleaved.py --what voxel --variant "old:LIB=$OLD" --variant new --variant "old2:LIB=$OLD" --rounds 20 --json $O/r05_ab_voxel.json 2>&1 | tail -5 ;;
  abrollout) timeout 900 python tools/ab_interleaved.py --what rollout --n-steps 32 --variant "old:LIB=$OLD" --variant new --variant "old2:LIB=$OLD" --rounds 8 --json $O/r05_ab_rollout.json 2>&1 | tail -5 ;;
  capstates) timeout 600 python tools/capture_states.py > $O/r05_capture_states.txt 2>$O/r05_capture_states.err; cat $O/r05_capture_states.txt ;;
  capone)    timeout 600 python tools/capture_states.py --one-stream --k 5 > $O/r05_capture_states_one_stream.txt 2>$O/r05_capture_states_one_stream.err; cat $O/r05_capture_states_one_stream.txt ;;
  tests_np2) GENNBV_WGRAD_NP=2 timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_ppo_gpu.py -m gpu -q --maxfail=6 -p no:cacheprovider > $O/r5_tests_np2.log 2>&1; tail -6 $O/r5_tests_np2.log ;;
  tests_enc) timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_ppo_gpu.py tests/test_ppo_g64_gpu.py -m gpu -q --maxfail=6 -p no:cacheprovider > $O/r5_tests_enc.log 2>&1; tail -6 $O/r5_tests_enc.log ;;
  abnp2)     timeout 1500 python tools/ab_interleaved.py --what train --captures 2 --variant "two:GENNBV_TAIL_MERGE=0" --variant sc1 --variant "np2:GENNBV_TAIL_MERGE=0,GENNBV_WGRAD_NP=2" --rounds 8 --json $O/r05_ab_train_sc1_np2.json 2>&1 | grep -v "^\[ab\]" | tail -14 ;;
  convnp2)   cd /tmp && export TMPDIR=/tmp; for v in 4 2; do rm -rf /tmp/prof_c; GENNBV_WGRAD_NP=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_c -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py > /tmp/prof_c.log 2>&1; echo "== GENNBV_WGRAD_NP=$v"; tail -1 /tmp/prof_c.log; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_c | grep -E "wgrad|reduce|finish|dgrad|conv12" | cut -c1-150; done | tee $O/r05_conv_np2_trace.txt; cd $GRAFT_REPO_ROOT ;;
  convpmc)   bash tools/conv_stall_pmc.sh > /dev/null 2>&1; wc -l $O/conv_stall_pmc.txt ;;
  *) echo "unknown stage $st" ;;
esac
done
