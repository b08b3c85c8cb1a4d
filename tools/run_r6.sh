#!/bin/bash
# Run on the GPU box: the measurement stages of round 6.   tools/run_r6.sh <stage> [<stage> ...]   (outputs under gpurun_out/)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for st in "$@"; do
case $st in
  probe)     timeout 120 tools/ubench/bin/lds_dma_probe > $O/r06_lds_dma_probe.txt 2>&1; cat $O/r06_lds_dma_probe.txt ;;
  t_dma)     timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -q -x -k "wgrad_lds_dma" -p no:cacheprovider 2>&1 | tail -5
             GENNBV_WGRAD_DMA=1 timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -q --maxfail=4 -k "forward_backward_vs_torch or fused_dgrad" -p no:cacheprovider 2>&1 | tail -5 ;;
  convdma)   cd /tmp && export TMPDIR=/tmp; for v in 0 1; do rm -rf /tmp/prof_c; GENNBV_WGRAD_DMA=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_c -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py > /tmp/prof_c.log 2>&1; echo "== GENNBV_WGRAD_DMA=$v"; tail -1 /tmp/prof_c.log; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_c | grep -E "wgrad|reduce|finish|dgrad|conv12" | cut -c1-150; done | tee $O/r06_conv_wgrad_dma_trace.txt; cd $GRAFT_REPO_ROOT ;;
  abdma)     timeout 1500 python tools/ab_interleaved.py --what train --captures 3 --variant base --variant "dma:GENNBV_WGRAD_DMA=1" --rounds 8 --json $O/r06_ab_train_wgrad_dma.json 2>&1 | grep -v "^\[ab\]" | tail -12 ;;
  abl)       cd /tmp && export TMPDIR=/tmp; for v in ${ABL_LIBS:-"" _abl_NOCONV _abl_NOCOMP _abl_NODMA _abl_NOCONV_NOCOMP}; do rm -rf /tmp/prof_c; GENNBV_HIP_LIB=$GRAFT_REPO_ROOT/gennbv_amd/libgennbv_hip$v.so GENNBV_WGRAD_DMA=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py > /tmp/prof_c.log 2>&1; echo "== lib$v"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_c | grep -E "wgrad_split|dgrad_c1w|conv12" | cut -c1-150; done | tee $O/${ABL_OUT:-r06_wgrad_dma_ablation.txt}; cd $GRAFT_REPO_ROOT ;;
  config5)   timeout 1500 python bench.py --steps 1 --warmup 1 --envs 512 --grid 128 --no-cpu-baseline --no-flat-rows 2>$O/r6_config5.err | tail -1 > $O/r06_bench_config5_shard_n1.json; python - <<PY
import json
d=json.load(open("$O/r06_bench_config5_shard_n1.json")); print("config5", round(d["value"]), "env-steps/s", d["breakdown_ms_per_step"], d["train_roofline"]["ms_per_minibatch"], d.get("timed_state_check"), d.get("encoder_roofline",{}).get("ms"))
PY
             ;;
  t_g128)    timeout 900 python -m pytest tests/test_ppo_g64_gpu.py -m gpu -q -x -k "g128" -p no:cacheprovider 2>&1 | tail -3 ;;
  wgpmc)     OUT=$O/r06_wgrad_pmc.txt; : > $OUT
             for v in 0 1; do echo "### GENNBV_WGRAD_DMA=$v" >> $OUT
               for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
                        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
                        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH" \
                        "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
                        "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
                        "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum" \
                        "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
                 echo "## --pmc $C" >> $OUT
                 GENNBV_WGRAD_DMA=$v $GRAFT_REPO_ROOT/tools/run_pmc.sh $OUT "k_conv2_wgrad_split" "$C" -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py --iters 3
                 grep -iE "error|invalid|not supported|unable" /tmp/pmc_run.log | head -2 >> $OUT
               done; done; cat $OUT | cut -c1-220 | tail -80 ;;
  dp200)     GENNBV_DP_SETTLE=0 PROBE_REPEAT=100 PROBE_GRAPH=1 timeout 1500 python tools/dp_probe.py > $O/r06_dp_200_captures_settle0.txt 2>&1; tail -4 $O/r06_dp_200_captures_settle0.txt; grep -c " ok " $O/r06_dp_200_captures_settle0.txt ;;
  dp200s)    PROBE_REPEAT=100 PROBE_GRAPH=1 timeout 1800 python tools/dp_probe.py > $O/r06_dp_200_captures.txt 2>&1; tail -2 $O/r06_dp_200_captures.txt; grep -c " ok " $O/r06_dp_200_captures.txt ;;
  t_dp)      timeout 1500 python -m pytest tests/test_parallel_gpu.py tests/test_ppo_gpu.py -m gpu -q --maxfail=6 --durations=6 -k "multi_rank or data_parallel or recapture" -p no:cacheprovider > $O/r6_tests_dp.log 2>&1; tail -15 $O/r6_tests_dp.log ;;
  dp1)       GENNBV_FORCE_DP=1 GENNBV_FORCE_SHARD=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-flat-rows 2>$O/r6_dp1.err | tail -1 > $O/r06_bench_dp1_n1.json; python - <<PY
import json
d=json.load(open("$O/r06_bench_dp1_n1.json")); print("dp1", round(d["ms_per_step"],1), "ms", d["train_roofline"]["ms_per_minibatch"], d.get("timed_state_check"), d["config"].get("dp_graph_mode"), d.get("dp_exchange_probe"))
PY
             ;;
  refdef)    timeout 1200 python bench.py --steps 3 --warmup 1 --height 400 --width 400 --grid 20 --no-flat-rows 2>$O/r6_refdef.err | tail -1 > $O/r06_bench_refdefault_n1.json; cut -c1-300 $O/r06_bench_refdefault_n1.json ;;
  semantic)  timeout 900 python bench.py --steps 3 --warmup 1 --semantic --no-cpu-baseline --no-flat-rows 2>/dev/null | tail -1 > $O/r06_bench_semantic_n1.json; cut -c1-300 $O/r06_bench_semantic_n1.json ;;
  t_small)   timeout 900 python -m pytest tests/test_ppo_gpu.py tests/test_encoder_gpu.py -m gpu -q -x -p no:cacheprovider > $O/r6_t_small.log 2>&1; tail -5 $O/r6_t_small.log ;;
  ab20)      timeout 900 python tools/ab_interleaved.py --what train --grid 20 --height 400 --width 400 --rounds 12 --captures 2 --variant "base:LIB=gennbv_amd/libgennbv_hip_r6base.so" --variant "new" --json $O/r06_ab_train_g20_b.json 2>&1 | tail -12 ;;
  ab64)      timeout 900 python tools/ab_interleaved.py --what train --rounds 12 --captures 3 --variant "base:LIB=gennbv_amd/libgennbv_hip_r6base.so" --variant "new" --json $O/r06_ab_train_g64_b.json 2>&1 | tail -12 ;;
  t_pipe)    timeout 900 python -m pytest tests/test_ppo_gpu.py -m gpu -q -x -p no:cacheprovider -k "pipelined or fused_train or recapture" > $O/r6_t_pipe.log 2>&1; tail -5 $O/r6_t_pipe.log ;;
  ab64p)     timeout 900 python tools/ab_interleaved.py --what train --rounds 12 --captures 3 --variant "onelaunch:GENNBV_PIPELINED_ADAM=0" --variant "pipelined" --json $O/r06_ab_train_g64_pipelined_adam.json 2>&1 | tail -12 ;;
  ab20p)     timeout 900 python tools/ab_interleaved.py --what train --grid 20 --height 400 --width 400 --rounds 12 --captures 2 --variant "onelaunch:GENNBV_PIPELINED_ADAM=0" --variant "pipelined" --json $O/r06_ab_train_g20_pipelined_adam.json 2>&1 | tail -10 ;;
  t_ppo)     timeout 900 python -m pytest tests/test_ppo_gpu.py -m gpu -q -x -p no:cacheprovider > $O/r6_t_ppo.log 2>&1; tail -5 $O/r6_t_ppo.log ;;
  ab64k)     timeout 900 python tools/ab_interleaved.py --what train --rounds 12 --captures 3 --variant "generic:GENNBV_PPO_GENERIC=1" --variant "sixheads" --json $O/r06_ab_train_g64_ppo_six_heads.json 2>&1 | tail -12 ;;
  ab20k)     timeout 900 python tools/ab_interleaved.py --what train --grid 20 --height 400 --width 400 --rounds 12 --captures 2 --variant "generic:GENNBV_PPO_GENERIC=1" --variant "sixheads" --json $O/r06_ab_train_g20_ppo_six_heads.json 2>&1 | tail -10 ;;
  ab64t)     timeout 900 python tools/ab_interleaved.py --what train --rounds 12 --captures 3 --variant "inline:GENNBV_WGRAD_TAIL_ASIDE=0" --variant "aside" --json $O/r06_ab_train_g64_wgrad_tail_aside.json 2>&1 | tail -12 ;;
  ab20t)     timeout 900 python tools/ab_interleaved.py --what train --grid 20 --height 400 --width 400 --rounds 12 --captures 2 --variant "inline:GENNBV_WGRAD_TAIL_ASIDE=0" --variant "aside" --json $O/r06_ab_train_g20_wgrad_tail_aside.json 2>&1 | tail -10 ;;
  ab64h)     timeout 900 python tools/ab_interleaved.py --what train --rounds 12 --captures 3 --variant "base:LIB=gennbv_amd/libgennbv_hip_r6head.so" --variant "new" --json $O/r06_ab_train_g64_bn2_means.json 2>&1 | tail -12 ;;
  ab20h)     timeout 900 python tools/ab_interleaved.py --what train --grid 20 --height 400 --width 400 --rounds 12 --captures 2 --variant "base:LIB=gennbv_amd/libgennbv_hip_r6head.so" --variant "new" --json $O/r06_ab_train_g20_bn2_means.json 2>&1 | tail -10 ;;
  ab64c)     timeout 900 python tools/ab_interleaved.py --what train --rounds 10 --captures 2 --variant "c128" --variant "c256:GENNBV_LIN_CHUNKS=256" --variant "c64:GENNBV_LIN_CHUNKS=64" --json $O/r06_ab_train_g64_lin_chunks.json 2>&1 | tail -12 ;;
  ab20r)     timeout 900 python tools/ab_interleaved.py --what rollout --grid 20 --height 400 --width 400 --rounds 30 --variant "general:GENNBV_ROLLOUT_PLAN=0" --variant "plan" --json $O/r06_ab_rollout_g20_flat_plan.json 2>&1 | tail -6 ;;
  ab20w)     timeout 900 python tools/ab_interleaved.py --what train --grid 20 --height 400 --width 400 --rounds 10 --captures 1 --variant "per8" --variant "per4:GENNBV_ADAM_PER=4" --variant "per16:GENNBV_ADAM_PER=16" --variant "per32:GENNBV_ADAM_PER=32" --json $O/r06_ab_train_g20_adam_per.json 2>&1 | tail -5 ;;
  tests)     timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 -p no:cacheprovider > $O/r6_tests.log 2>&1; tail -30 $O/r6_tests.log ;;
  bench)     timeout 900 python bench.py --steps 5 --warmup 2 2>$O/r6_bench.err | tail -1 > $O/r6_bench_n1.json; cut -c1-900 $O/r6_bench_n1.json ;;
  benchdrv)  timeout 1200 python bench.py --steps 20 --warmup 5 2>$O/r6_benchdrv.err | tail -1 > $O/r6_bench_driver_cfg_n1.json; cut -c1-600 $O/r6_bench_driver_cfg_n1.json ;;
  *) echo "unknown stage $st" ;;
esac
done
