"""A/B: the captured minibatch as one stream (no pose-branch fork, no second-stream weight gradient) vs the default two streams.
    python tools/ab_streams.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

for mode in ("default", "no_async_wgrad", "single_stream"):
    args = bench.parse()
    args.steps, args.warmup, args.n_steps = 2, 1, 32
    algo, cfg, env = bench.build_algo(args, "cuda:0", 0, 1)
    if mode in ("no_async_wgrad", "single_stream"):
        algo.async_wgrad = False
    if mode == "single_stream":
        algo.policy.features_extractor.overlap_branches = False
    algo._setup_learn(total_timesteps=10 ** 12)
    ph = {"rollout": bench.Phase(), "train": bench.Phase()}
    bench.one_iteration(algo, {"rollout": bench.Phase(), "train": bench.Phase()})
    torch.cuda.synchronize()
    for _ in range(args.steps):
        bench.one_iteration(algo, ph)
    torch.cuda.synchronize()
    n_mb = args.envs * args.n_steps // args.batch_size * args.n_epochs
    print(f"{mode:16s} train {ph['train'].total_ms() / args.steps:8.1f} ms / iteration = {1e3 * ph['train'].total_ms() / args.steps / n_mb:6.1f} us per minibatch")
    del algo, env
    torch.cuda.empty_cache()
