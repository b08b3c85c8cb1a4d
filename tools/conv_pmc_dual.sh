#!/bin/bash
# HBM traffic of the two split backward kernels as separate launches vs as the dual launch (B = 128, G = 64): FETCH_SIZE passes only
OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_pmc_dual.txt
: > $OUT
echo "# tools/conv_pmc_dual.sh: FETCH_SIZE in KB, to be doubled per MI355X_MICROARCH.md section HBM" >> $OUT
for D in 0 1; do
  echo "## GENNBV_BWD_DUAL=$D --pmc FETCH_SIZE" >> $OUT
  GENNBV_BWD_DUAL=$D $GRAFT_REPO_ROOT/tools/run_pmc.sh $OUT "k_conv2_wgrad_split|k_conv2_dgrad_c1w_split|k_conv2_bwd_dual_split|k_conv12_fwd" "FETCH_SIZE" -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py --iters 3
done
cat $OUT
