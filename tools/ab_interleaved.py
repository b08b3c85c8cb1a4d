"""Interleaved A/B of kernel variants inside ONE process on ONE box.

Round 3's A/Bs ran whole bench processes one after the other: +-3 % box-to-box, and a replayed graph that lands in one of two
states per PROCESS (561-566 vs 581-589 us) -- 10 us effects did not resolve.  Here every variant is built and captured once, in the
same process, and the variants are then timed alternately in short chunks, many rounds, so that clock / thermal drift is common to
all of them:

    python tools/ab_interleaved.py --what train  --variant base --variant "nofuse:GENNBV_FUSED_TRAIN=0" --variant "old:LIB=/path/libold.so"
    python tools/ab_interleaved.py --what voxel  --variant base --variant "old:LIB=gennbv_amd/libgennbv_hip_old.so"
    python tools/ab_interleaved.py --what rollout ...

A variant is `name[:K=V,K=V,...]`; `LIB=path` selects another build of libgennbv_hip.so (gennbv_amd._lib.activate), `STREAM=hp` builds,
captures and replays the variant on a high-priority HIP stream (the streams the code forks onto keep the default priority), everything else
is an environment variable that is set while the variant is BUILT and CAPTURED (the library reads its switches at call / capture
time; a replayed hipGraph has them baked in).  `train`: the captured PPO minibatch graph of bench.py's algorithm object (learning
rate 1e-12: thousands of replays must not walk the parameters away).  `voxel`: gnbv_update_occ_grid_coded as the env calls it
(eager launches).  `rollout`: one env step + policy forward of collect_rollouts (eager; or its graph when the build has one).

Output per variant: median and MAD of the per-call time over all chunks, and -- against the FIRST variant -- the median of the
per-round paired differences with its MAD: the number to quote."""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_variant(v: str):
    name, _, rest = v.partition(":")
    env, lib = {}, None
    for kv in filter(None, rest.split(",")):
        k, _, val = kv.partition("=")
        if k == "LIB":
            lib = val
        else:
            env[k] = val  # (STREAM stays in here: main() pops it)
    return name, env, lib


class _OnStream:
    """everything inside runs with `stream` current (None: the ambient stream), ordered behind what was enqueued before"""
    def __init__(self, stream):
        self.stream, self.ctx = stream, None

    def __enter__(self):
        if self.stream is not None:
            import torch
            self.stream.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            import torch
            self.ctx.__exit__(*a)
            torch.cuda.current_stream().wait_stream(self.stream)


class _Env:
    def __init__(self, env):
        self.env, self.old = env, {}

    def __enter__(self):
        for k, v in self.env.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def build_train(a, dev):
    import torch
    import bench
    ns = argparse.Namespace(gpus=1, steps=1, warmup=0, envs=a.envs, grid=a.grid, height=a.height, width=a.width, n_steps=a.n_steps, batch_size=a.batch_size,
                            n_epochs=16, frames=2, backend="hip", obs="compact", target_kl="off", semantic=a.semantic, no_cpu_baseline=True,
                            no_flat_rows=True, no_state_check=True)
    algo, cfg, env = bench.build_algo(ns, dev, 0, 1)
    algo.learning_rate = 1e-12
    algo.lr_schedule = lambda _: 1e-12
    algo.graph_candidates = 1  # (every harness capture is ONE capture: the spread over captures is what --captures measures)
    algo._setup_learn(total_timesteps=10 ** 12)
    algo.collect_rollouts(algo.env, None, algo.rollout_buffer, n_rollout_steps=algo.n_steps)
    algo.train()
    g = algo._hip["graph"]
    assert g is not None and not isinstance(g, tuple)
    torch.cuda.synchronize()
    loss = algo._hip["loss"]
    # the same capture with the update MASKED (what PPO_Grid_Obs._best_of_captures ranks its candidates by): printed beside the real figure
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    loss.stop_flag.fill_(1)
    for j in range(33):
        if j == 3:
            e0.record()
        loss.stats_row.zero_()
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"[ab]   masked replay of this capture: {e0.elapsed_time(e1) / 30.0 * 1e3:.2f} us", file=sys.stderr, flush=True)
    loss.stop_flag.zero_()
    loss.stats_row.zero_()
    assert loss.stats.shape[0] > 64  # (every replay appends a row to the statistics table: the chunk must fit, `reset` rewinds it)
    return {"fn": (lambda: g.replay()), "reset": (lambda: loss.stats_row.zero_()), "max_chunk": int(loss.stats.shape[0]) - 2}, algo


def build_voxel(a, dev):
    import torch
    from gennbv_amd.env import synthetic as S
    from gennbv_amd.env.config import TaskConfig
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    cfg = TaskConfig(camera_width=a.width, camera_height=a.height, grid_size=a.grid)
    scene = S.make_scenes(a.envs, a.grid, seed=1, device=dev)
    frames = S.make_frames(scene, cfg, 4, seed=1, with_rgba=False)
    upd = OccupancyGridUpdater(a.envs, a.grid, a.height, a.width, S.inverse_intrinsics(a.height, a.width), scene.range_gt, scene.voxel_size, scene.grid_gt,
                               dev, max_steps_between_resets=100)
    t8 = torch.zeros(a.envs, a.grid ** 3, dtype=torch.int8, device=dev)
    c2ws = [S.c2w_from_view(f.view, scene.env_origins) for f in frames]
    poses = [f.poses.contiguous() for f in frames]
    all_reset = torch.ones(a.envs, dtype=torch.uint8, device=dev)
    state = {"i": 0}

    def step():
        i = state["i"] = state["i"] + 1
        k = i % 4
        upd.update(frames[k].depth_raw, frames[k].seg_raw, c2ws[k], poses[k], reset_mask=all_reset if i % 64 == 63 else None, tri_i8_out=t8, fp32_out=False)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    return step, (upd, frames, t8)


def build_rollout(a, dev):
    import torch
    import bench
    ns = argparse.Namespace(gpus=1, steps=1, warmup=0, envs=a.envs, grid=a.grid, height=a.height, width=a.width, n_steps=a.n_steps, batch_size=a.batch_size,
                            n_epochs=1, frames=4, backend="hip", obs="compact", target_kl="off", semantic=a.semantic, no_cpu_baseline=True,
                            no_flat_rows=True, no_state_check=True)
    algo, cfg, env = bench.build_algo(ns, dev, 0, 1)
    algo._setup_learn(total_timesteps=10 ** 12)

    vox = bench.instrument_voxel(algo.env)  # HIP events around every gnbv_update_occ_grid call INSIDE the rollout (what bench.py's roofline prices)

    def step():  # one whole rollout of n_steps env steps (per-call time is divided by n_steps below)
        algo.collect_rollouts(algo.env, None, algo.rollout_buffer, n_rollout_steps=algo.n_steps)
    step()
    torch.cuda.synchronize()
    return {"fn": step, "vox": vox}, algo


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="train", choices=["train", "voxel", "rollout"])
    ap.add_argument("--variant", action="append", required=True)
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--captures", type=int, default=1, help="independent builds / graph captures per variant: a replayed hipGraph lands in one of "
                    "several states per CAPTURE (+-12 us per minibatch between two captures of the same library, +-0.4 us inside one), so a "
                    "kernel change is only resolved against the spread over captures")
    ap.add_argument("--chunk", type=int, default=None, help="calls per timed chunk (default: 50 train / 50 voxel / 1 rollout)")
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--n-steps", type=int, default=8)
    ap.add_argument("--batch-size", type=int, default=128)
    ap.add_argument("--semantic", action="store_true")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import torch
    from gennbv_amd import _lib
    dev = "cuda:0"
    torch.cuda.set_device(dev)
    build = {"train": build_train, "voxel": build_voxel, "rollout": build_rollout}[a.what]
    chunk = a.chunk or {"train": 50, "voxel": 50, "rollout": 1}[a.what]
    per_call_div = a.n_steps if a.what == "rollout" else 1
    variants = []
    for v in [x for _ in range(a.captures) for x in a.variant]:
        name, env, lib = parse_variant(v)
        stream = torch.cuda.Stream(priority=-1) if env.pop("STREAM", None) == "hp" else None
        print(f"[ab] building variant {name} env={env} lib={lib} stream={'high priority' if stream else 'ambient'}", file=sys.stderr, flush=True)
        with _Env(env), _OnStream(stream):
            _lib.activate(lib)
            fn, keep = build(a, dev)
        if not isinstance(fn, dict):
            fn = {"fn": fn}
        if a.what != "train":  # eager launches read the library's switches at CALL time: the variant's environment / library around every chunk
            fn["chunk_ctx"] = (env, lib)
        torch.cuda.synchronize()
        print(f"[ab] variant {name} built and captured", file=sys.stderr, flush=True)
        variants.append({"name": name, "env": env, "lib": lib, "fn": fn["fn"], "reset": fn.get("reset"), "ctx": fn.get("chunk_ctx"), "keep": keep, "t": [], "stream": stream,
                         "vox": fn.get("vox"), "tv": []})
        if fn.get("max_chunk"):
            chunk = min(chunk, fn["max_chunk"])
    _lib.activate(None)
    # GPU-busy warm-up (leave the idle power state: 0.3 s of the first variant), then the alternating rounds
    t0 = time.perf_counter()
    def run_chunk(v, timed: bool):
        env, lib = v["ctx"] if v["ctx"] else ({}, None)
        with _Env(env if v["ctx"] else {}), _OnStream(v["stream"]):
            if v["ctx"]:
                _lib.activate(lib)
            if v["reset"]:
                v["reset"]()
            v["fn"]()  # one untimed call: the variant's code / data back in the caches
            if timed:
                ev0.record()
            for _ in range(chunk):
                v["fn"]()
            if timed:
                ev1.record()
            torch.cuda.synchronize()
            if v.get("vox") is not None:
                ph = v["vox"]
                if timed and ph.count():
                    v["tv"].append(ph.total_ms() / ph.count() * 1e3)
                ph.pairs.clear()
        return ev0.elapsed_time(ev1) * 1e3 / chunk / per_call_div if timed else None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.perf_counter() - t0 < 0.3:
        run_chunk(variants[0], False)
    for r in range(a.rounds):
        if r < 2:
            print(f"[ab] round {r}", file=sys.stderr, flush=True)
        order = variants[r % len(variants):] + variants[:r % len(variants)]  # rotate who goes first
        for v in order:
            v["t"].append(run_chunk(v, True))  # us per call
    unit = {"train": "us / minibatch", "voxel": "us / update", "rollout": "us / env step"}[a.what]
    base = variants[0]
    out = {"what": a.what, "unit": unit, "rounds": a.rounds, "chunk": chunk, "device": torch.cuda.get_device_name(0), "variants": []}
    if a.captures > 1:  # pool the captures of a variant: the spread over captures is the noise floor of a comparison
        names = []
        for v in variants:
            if v["name"] not in names:
                names.append(v["name"])
        print(f"# {a.captures} captures per variant; per-capture medians and the pooled figure:")
        pooled = []
        for nm in names:
            caps = [statistics.median(v["t"]) for v in variants if v["name"] == nm]
            allt = [x for v in variants if v["name"] == nm for x in v["t"]]
            print(f"#   {nm:18s} captures: " + " ".join(f"{c:8.2f}" for c in caps) + f"   mean of captures {sum(caps) / len(caps):8.2f}, spread {max(caps) - min(caps):5.2f}")
            pooled.append({"name": nm, "capture_medians": caps, "mean_of_captures": sum(caps) / len(caps), "min_capture": min(caps), "max_capture": max(caps),
                           "median_all": statistics.median(allt)})
        out["pooled"] = pooled
    for v in variants:
        med = statistics.median(v["t"])
        mad = statistics.median(abs(x - med) for x in v["t"])
        diffs = [x - y for x, y in zip(v["t"], base["t"])]
        dmed = statistics.median(diffs)
        dmad = statistics.median(abs(x - dmed) for x in diffs)
        out["variants"].append({"name": v["name"], "env": v["env"], "lib": v["lib"], "median": med, "mad": mad, "min": min(v["t"]), "max": max(v["t"]),
                                "paired_delta_vs_first": dmed, "paired_delta_mad": dmad})
        print(f"{v['name']:20s} {med:9.2f} +- {mad:5.2f} {unit}   [min {min(v['t']):.2f}, max {max(v['t']):.2f}]   vs {base['name']}: {dmed:+7.2f} +- {dmad:4.2f}")
    if any(v["tv"] for v in variants):
        print("# voxel update inside the rollout (HIP events around the three launches), us per update:")
        out["voxel_in_rollout_us"] = []
        for v in variants:
            med = statistics.median(v["tv"])
            diffs = [x - y for x, y in zip(v["tv"], base["tv"])]
            out["voxel_in_rollout_us"].append({"name": v["name"], "median": med, "min": min(v["tv"]), "max": max(v["tv"]), "paired_delta_vs_first": statistics.median(diffs)})
            print(f"#   {v['name']:18s} {med:8.2f}   [min {min(v['tv']):.2f}, max {max(v['tv']):.2f}]   vs {base['name']}: {statistics.median(diffs):+6.2f}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
