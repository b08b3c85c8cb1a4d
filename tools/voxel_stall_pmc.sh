#!/bin/bash
# Stall-reason counters of the three voxel-update kernels (256 envs x 240x320 x 64^3): separate rocprofv3 --pmc passes (--kernel-trace
# only) over tools/microbench_voxel.py.  Writes gpurun_out/voxel_stall_pmc.txt (copy to profiles/rNN_voxel_stall_pmc.txt).
OUT=$GRAFT_REPO_ROOT/gpurun_out/voxel_stall_pmc.txt
RX="k_hit_list|k_ray_list|k_grid_update_coded"
: > $OUT
echo "# tools/voxel_stall_pmc.sh: per-kernel averages over the dispatches of tools/microbench_voxel.py --iters 5" >> $OUT
for C in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  echo "## --pmc $C" >> $OUT
  $GRAFT_REPO_ROOT/tools/run_pmc.sh $OUT "$RX" "$C" -- python $GRAFT_REPO_ROOT/tools/microbench_voxel.py --iters 5
  grep -iE "error|invalid|not supported|unable" /tmp/pmc_run.log | head -3 >> $OUT
done
cat $OUT
