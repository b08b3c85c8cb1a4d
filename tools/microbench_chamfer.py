"""Time gnbv_chamfer_distance at evaluation scale (scanned cloud n vs GT cloud m)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gennbv_amd.eval import metrics as M

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--m", type=int, default=100_000); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
x = (torch.rand(a.n, 3, device="cuda:0") - 0.5) * 16
y = (torch.rand(a.m, 3, device="cuda:0") - 0.5) * 16
M.chamfer_distance(x, y); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    d = M.chamfer_distance(x, y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
pairs = 2.0 * a.n * a.m
print(f"chamfer n={a.n} m={a.m}: {ms:.2f} ms  -> {pairs / ms / 1e9:.2f} T pair-evaluations/s ({7 * pairs / ms / 1e9:.0f} T lane-instructions/s at 7 VALU ops per pair), value {float(d):.6f}")
