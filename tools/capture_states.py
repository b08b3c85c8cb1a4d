"""Why does a captured minibatch land in a 507 or a 521-532 us state per CAPTURE?

Hypothesis tested here: every capture allocates its intermediates (y1 244 MB, y2 / dy2 / workspaces) from its OWN private memory
pool, i.e. at different device addresses, and the kernels' HBM channel / bank pattern depends on how those buffers lie relative to
each other.  Protocol, one algorithm object, one process:
  A  K captures, each DROPPED before the next is taken (the caching allocator hands the freed blocks out again: same addresses)
  B  K captures kept alive together (K different pools: different addresses)
per capture: the masked-replay time (stop_flag = 1, as PPO_Grid_Obs._best_of_captures ranks candidates) and the addresses of the
pool's large segments.    python tools/capture_states.py [--k 6]"""
import argparse
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=6)
ap.add_argument("--one-stream", action="store_true", help="no second stream inside the captured minibatch (pose branch and fc_grid's dW on the main stream)")
a = ap.parse_args()
dev = "cuda:0"
ns = argparse.Namespace(gpus=1, steps=1, warmup=0, envs=256, grid=64, height=240, width=320, n_steps=8, batch_size=128, n_epochs=16, frames=2,
                        backend="hip", obs="compact", target_kl="off", semantic=False, no_cpu_baseline=True, no_flat_rows=True, no_state_check=True)
algo, cfg, env = bench.build_algo(ns, dev, 0, 1)
algo.learning_rate = 1e-12
algo.lr_schedule = lambda _: 1e-12
algo.graph_candidates = 1
if a.one_stream:
    algo.async_wgrad = False
    algo.policy.features_extractor.overlap_branches = False
algo._setup_learn(total_timesteps=10 ** 12)
algo.collect_rollouts(algo.env, None, algo.rollout_buffer, n_rollout_steps=algo.n_steps)
algo.train()
st = algo._hip
loss = st["loss"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def big_segments(before):
    segs = [(s["address"], s["total_size"]) for s in torch.cuda.memory_snapshot() if s["total_size"] >= 16 << 20]
    return sorted(set(segs) - before), set(segs)


def capture():
    loss.stop_flag.fill_(1)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        algo._hip_minibatch_body(st)
    return g


def timed(g, reps=3):
    ts = []
    for _ in range(reps):
        loss.stop_flag.fill_(1)
        for j in range(33):
            if j == 3:
                e0.record()
            loss.stats_row.zero_()
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30.0 * 1e3)
    return sorted(ts)[len(ts) // 2]


st["graph"] = None
gc.collect()
torch.cuda.empty_cache()
print(f"# one_stream={a.one_stream}")
print("# A: sequential captures, each dropped before the next")
for i in range(a.k):
    _, before = big_segments(set())
    g = capture()
    new, _ = big_segments(before)
    t = timed(g)
    print(f"A{i}: {t:7.2f} us   new segments: " + " ".join(f"{ad:#x}+{sz >> 20}M" for ad, sz in new))
    del g
    gc.collect()
print("# B: captures kept alive together")
keep = []
for i in range(a.k):
    _, before = big_segments(set())
    g = capture()
    new, _ = big_segments(before)
    keep.append((g, new))
for rnd in range(2):
    for i, (g, new) in enumerate(keep):
        t = timed(g)
        print(f"B{i} (round {rnd}): {t:7.2f} us   segments: " + " ".join(f"{ad:#x}+{sz >> 20}M" for ad, sz in new))
