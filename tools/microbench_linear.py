"""Time the split-K fc_grid kernel against the library GEMM (torch.nn.functional.linear) at M x N x K."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gennbv_amd.ops.encoder_ops import linear_relu

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=128); ap.add_argument("--n", type=int, default=256); ap.add_argument("--k", type=int, default=54000)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = "cuda:0"
x = torch.rand(a.m, a.k, device=dev); lin = torch.nn.Linear(a.k, a.n).to(dev)


def timeit(f):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


with torch.no_grad():
    t_hip = timeit(lambda: linear_relu(x, lin))
    t_lib = timeit(lambda: torch.relu(torch.nn.functional.linear(x, lin.weight, lin.bias)))
print(f"M={a.m} N={a.n} K={a.k}: split-K HIP {t_hip:.1f} us, library GEMM + relu {t_lib:.1f} us")
