"""Where does the HOST time of a rollout step go?  cProfile over one collect_rollouts() of bench.py's algorithm object (the GPU is
never waited for inside the rollout, so the profile is pure host time: Python + ctypes + launches)."""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

_ap = argparse.ArgumentParser()
_ap.add_argument("--grid", type=int, default=64)
_ap.add_argument("--height", type=int, default=240)
_ap.add_argument("--width", type=int, default=320)
_a = _ap.parse_args()
ns = argparse.Namespace(gpus=1, steps=1, warmup=0, envs=256, grid=_a.grid, height=_a.height, width=_a.width, n_steps=64, batch_size=128, n_epochs=1, frames=4,
                        backend="hip", obs="compact", target_kl="off", semantic=False, no_cpu_baseline=True, 
                        no_flat_rows=True, no_state_check=True)
algo, cfg, env = bench.build_algo(ns, "cuda:0", 0, 1)
algo._setup_learn(total_timesteps=10 ** 12)
for _ in range(2):
    algo.collect_rollouts(algo.env, None, algo.rollout_buffer, n_rollout_steps=algo.n_steps)
torch.cuda.synchronize()
t0 = time.perf_counter()
algo.collect_rollouts(algo.env, None, algo.rollout_buffer, n_rollout_steps=algo.n_steps)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per env step {1e6 * (t1 - t0) / ns.n_steps:.1f} us (enqueue only), wall incl. GPU drain {1e6 * (t2 - t0) / ns.n_steps:.1f} us")
pr = cProfile.Profile()
pr.enable()
algo.collect_rollouts(algo.env, None, algo.rollout_buffer, n_rollout_steps=algo.n_steps)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
