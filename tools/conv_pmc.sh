#!/bin/bash
# HBM traffic of the conv-stack kernels of one minibatch (B = 128, G = 64): separate rocprofv3 --pmc passes over
# tools/microbench_conv.py.  Writes gpurun_out/conv_pmc.txt (copy to profiles/rNN_conv_pmc.txt).
OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_pmc.txt
: > $OUT
echo "# tools/conv_pmc.sh: per-kernel averages, B = 128, G = 64; FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE to be doubled per MI355X_MICROARCH.md section HBM)" >> $OUT
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  echo "## --pmc $C" >> $OUT
  $GRAFT_REPO_ROOT/tools/run_pmc.sh $OUT "k_conv1_fwd_split|k_conv2_fwd_split|k_conv2_wgrad_split|k_conv2_dgrad_c1w_split" "$C" -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py --iters 3
done
cat $OUT
