cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ppo_gpu.py tests/test_range_guard_gpu.py tests/test_ppo_g64_gpu.py "tests/test_encoder_gpu.py::test_bn2_relu_folded_into_fc_grid_is_bit_identical" "tests/test_encoder_gpu.py::test_fc_grid_forward_applies_the_owed_adam_update_bit_identically" -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
for f in 0 1 0 1; do
echo "FUSE=$f"; GENNBV_ADAM_FUSE=$f python bench.py --no-cpu-baseline --no-flat-rows --no-state-check 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
bash tools/prof_minibatch.sh r03g
