#!/bin/bash
# Stall-reason counters of the split conv kernels of one minibatch (B = 128, G = 64): separate rocprofv3 --pmc passes (8 SQ slots each,
# --kernel-trace only) over tools/microbench_conv.py.  Writes gpurun_out/conv_stall_pmc.txt (copy to profiles/rNN_conv_pmc.txt).
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md): WAIT_ANY (parked on s_waitcnt / barrier)
# + WAIT_INST_ANY (ready, not issued) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES.
OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_stall_pmc.txt
RX="k_conv12_fwd_split|k_conv2_wgrad_split|k_conv2_dgrad_c1w_split"
: > $OUT
echo "# tools/conv_stall_pmc.sh: per-kernel averages over the dispatches of tools/microbench_conv.py --iters 3 (B = 128, G = 64)" >> $OUT
for C in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_CVT" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
  "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum"; do
  echo "## --pmc $C" >> $OUT
  $GRAFT_REPO_ROOT/tools/run_pmc.sh $OUT "$RX" "$C" -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py --iters 3
  grep -iE "error|invalid|not supported|unable" /tmp/pmc_run.log | head -3 >> $OUT
done
cat $OUT
