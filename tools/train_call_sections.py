"""Where does a train() call spend its time OUTSIDE the 1 280 minibatch replays?  bench.py's algorithm object (BASELINE configs[1]), two
iterations of warm-up, then one train() call with a device synchronisation and a wall-clock stamp around each of its sections
(_check_ranges / snapshot / per-call tables / the replay loop / the log read-back).  What bench.py reports as `ms_per_minibatch` is the
whole call divided by the replays: this shows how much of it is not a replay."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--height", type=int, default=240)
ap.add_argument("--width", type=int, default=320)
a = ap.parse_args()
ns = argparse.Namespace(gpus=1, steps=1, warmup=0, envs=256, grid=a.grid, height=a.height, width=a.width, n_steps=128, batch_size=128, n_epochs=5, frames=4,
                        backend="hip", obs="compact", target_kl="off", semantic=False, no_cpu_baseline=True, no_flat_rows=True, no_state_check=True)
algo, cfg, env = bench.build_algo(ns, "cuda:0", 0, 1)
algo._setup_learn(total_timesteps=10 ** 12)
for _ in range(2):
    algo.collect_rollouts(algo.env, None, algo.rollout_buffer, n_rollout_steps=algo.n_steps)
    algo.train()
algo.collect_rollouts(algo.env, None, algo.rollout_buffer, n_rollout_steps=algo.n_steps)
torch.cuda.synchronize()

sections = []


def timed(name, fn):
    def wrapper(*args, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*args, **kw)
        torch.cuda.synchronize()
        sections.append((name, 1e3 * (time.perf_counter() - t0)))
        return out
    return wrapper


for name in ("_check_ranges", "_snapshot_update_state", "_train_call_tables", "_train_call_run", "_train_call_log"):
    setattr(algo, name, timed(name, getattr(algo, name)))
enc = algo.policy.features_extractor
enc.check_operand_ranges = timed("enc.check_operand_ranges", enc.check_operand_ranges)
torch.cuda.synchronize()
t0 = time.perf_counter()
algo.train()
torch.cuda.synchronize()
total = 1e3 * (time.perf_counter() - t0)
for name, ms in sections:
    print(f"{name:28s} {ms:9.3f} ms")
print(f"{'train() call':28s} {total:9.3f} ms   (sections nest: _check_ranges contains enc.check_operand_ranges)")
