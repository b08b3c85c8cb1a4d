#!/bin/bash
# run on the GPU box: A/B of the PPO update under environment switches, alternating, same box.
# (round 3's harness, one process per configuration: differences under ~15 us per minibatch are inside its capture-to-capture noise --
#  tools/ab_interleaved.py, one process with every variant captured and replayed alternately, replaced it in round 4.)
#   tools/ab_train.sh "NAME=ENV1=a ENV2=b" "NAME2=..." ...   (each argument: label=space-separated env assignments)
cd $GRAFT_REPO_ROOT
REPS=${REPS:-2}
for r in $(seq 1 $REPS); do
  for v in "$@"; do
    label=${v%%=*}; envs=${v#*=}
    line=$(env $envs python bench.py --steps 2 --warmup 1 --n-steps ${NSTEPS:-32} --no-cpu-baseline --no-flat-rows --no-state-check 2>/dev/null | tail -1)
    python - "$label" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
c = d["config"]; b = c["breakdown_ms_per_step"]; n = c["minibatches_last_iteration"]
print(f"{sys.argv[1]:24s} train {b['train']:8.2f} ms = {b['train'] / n * 1e3:7.1f} us / minibatch ({n}),  rollout {b['rollout']:7.2f} ms, voxel launch {d['roofline']['launch_ms'] * 1e3:6.1f} us")
PY
  done
done
