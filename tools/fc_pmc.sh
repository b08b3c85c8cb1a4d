#!/bin/bash
# per-kernel trace + SQ counters of the fc_grid kernels (tools/microbench_fc.py).  Writes gpurun_out/fc_pmc.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/fc_pmc.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_fc
rocprofv3 --kernel-trace --stats -d /tmp/prof_fc -- python $GRAFT_REPO_ROOT/tools/microbench_fc.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/prof_fc | grep -E "^kernel|k_" | cut -c1-160 >> $OUT
RX="k_linear_splitk_split|k_skinny_gemm_split|k_fc_bwd_prep"
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  echo "## --pmc $C" >> $OUT
  $GRAFT_REPO_ROOT/tools/run_pmc.sh $OUT "$RX" "$C" -- python $GRAFT_REPO_ROOT/tools/microbench_fc.py --iters 4
  grep -i "error\|invalid\|not found" /tmp/pmc_run.log | head -3 >> $OUT
done
cat $OUT
