"""fc_grid (Linear 54000 -> 256 at 128 rows) forward + backward through the product path, in a loop, with a 1 GB pass between the
iterations (so that the 55 MB weight is not served from the memory-side cache): the workload of the per-kernel traces / PMC passes of
k_linear_splitk_split, k_fc_bwd_prep and k_skinny_gemm_split<2 | 4>.   python tools/microbench_fc.py [--iters 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gennbv_amd.ops.encoder_ops import linear_relu

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=128); ap.add_argument("--n", type=int, default=256); ap.add_argument("--k", type=int, default=54000)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
x = torch.rand(a.m, a.k, device=dev, requires_grad=True)
lin = torch.nn.Linear(a.k, a.n).to(dev)
lin.weight.grad, lin.bias.grad = torch.zeros_like(lin.weight), torch.zeros_like(lin.bias)
flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)
g = torch.randn(a.m, a.n, device=dev)
for _ in range(a.iters):
    flush.add_(1.0)
    y = linear_relu(x, lin)
    flush.add_(1.0)
    y.backward(g)
    x.grad = None
torch.cuda.synchronize()
print("done", float(y.sum()))
