"""Per-kernel times of the conv stack (forward + backward through the C-ABI) at a given batch: run under
`rocprofv3 --kernel-trace --stats` and read the averages, e.g. to compare an L2-resident batch (8) with the bench's 128."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gennbv_amd.network.hybrid_encoder import Hybrid_Encoder
from gennbv_amd.ops import encoder_ops
from gennbv_amd.spaces import Box
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rows", type=int, default=0, help="rows of the observation pool (default: 2 x batch)")
a = ap.parse_args()
dev, g, b = "cuda:0", a.grid, a.batch
n = a.rows or 2 * b
enc = Hybrid_Encoder(Box(-np.inf, np.inf, (600 + g ** 3 + 8192,)), encoder_param={}, net_param={"append_hidden_shapes": [256, 256]},
                     state_input_shape=(600,), visual_input_shape=(2, 64, 64), grid_size=g).to(dev)
gen = torch.Generator().manual_seed(0)
grid_i8 = (torch.randint(-1, 2, (n, g ** 3), generator=gen) * (torch.rand(n, g ** 3, generator=gen) < 0.4)).to(torch.int8).to(dev)
small = torch.randn(n, 600 + 8192, generator=gen).to(dev)
ac = encoder_ops.input_autocorr(grid_i8, g)
rows = torch.randperm(n)[:b].to(dev)
seq = enc.naive_encoder_grid
enc.train()
for it in range(a.iters + 3):
    if it == 3:
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    f = encoder_ops.grid_encoder(small, rows, 600, g, seq, True, grid_i8=grid_i8, compact=True, autocorr=ac)
    f.backward(torch.ones_like(f))
e1.record(); torch.cuda.synchronize()
print(f"batch {b} grid {g}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per conv-stack forward+backward")
