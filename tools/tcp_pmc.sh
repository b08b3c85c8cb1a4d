OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_tcp_pmc.txt
: > $OUT
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TA_BUSY_sum TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  echo "## --pmc $C" >> $OUT
  bash $GRAFT_REPO_ROOT/tools/run_pmc.sh $OUT "k_conv2_fwd|k_conv2_wgrad|k_conv2_dgrad_c1w|k_linear_splitk" "$C" -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py --batch 128 --iters 3
  tail -2 /tmp/pmc_run.log | cut -c1-200 >> $OUT.log
done
cat $OUT | head -120
