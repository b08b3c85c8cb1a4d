"""Write a synthetic recording as a GNBVFEED container and time loading it back into an HBM-resident pool.

    python tools/record_feed.py --out /tmp/feed.gnbv [--envs 256 --frames 8 --grid 64 --depth f16]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gennbv_amd.env import feed_file as FF, synthetic as S
from gennbv_amd.env.config import TaskConfig

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True); ap.add_argument("--envs", type=int, default=256); ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--grid", type=int, default=64); ap.add_argument("--depth", default="f16", choices=["f16", "f32"])
ap.add_argument("--device", default="cuda:0" if torch.cuda.is_available() else "cpu")
a = ap.parse_args()
cfg = TaskConfig(camera_width=320, camera_height=240, grid_size=a.grid)
scene = S.make_scenes(a.envs, a.grid, seed=1, device=a.device)
frames = S.make_frames(scene, cfg, a.frames, seed=1)
t0 = time.perf_counter()
FF.record(a.out, scene, frames, S.inverse_intrinsics(240, 320, cfg.horizontal_fov), depth_dtype=a.depth)
t1 = time.perf_counter()
size = os.path.getsize(a.out)
ff = FF.FeedFile(a.out)
feed = FF.load_feed(ff, a.device)
if a.device.startswith("cuda"):
    torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{a.out}: {size / 1e6:.1f} MB for {a.frames} frames x {a.envs} envs ({size / a.frames / a.envs / 1e3:.1f} KB per env-frame, depth {a.depth}); "
      f"write {size / 1e6 / (t1 - t0):.0f} MB/s, map + upload + decode {size / 1e6 / (t2 - t1):.0f} MB/s")
