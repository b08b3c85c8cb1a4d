"""What happened to BENCH_r03's encoder_roofline (4.57 ms against 0.386 ms for the same conv-stack step)?  Hypothesis: the figure was
taken right after seconds of CPU-only work (the oracle replay of timed_state_check), 3 + 20 iterations = 9 ms of GPU work in all,
i.e. entirely inside the device's ramp out of its idle power state.  This script measures that ramp: a fixed kernel sequence (the
conv-stack forward + backward of one minibatch) timed call by call (a) back to back after a long busy period, (b) right after N
seconds of GPU idleness.  Prints the first 40 per-call times of each phase and when the times settle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gennbv_amd.ops import encoder_ops
from tests import policy_util as pu

dev = "cuda:0"
g, b = 64, 128
pol, _, _ = pu.make_policy(g=g, device=dev, backend="hip", det_weights=False)
seq = pol.features_extractor.naive_encoder_grid
pol.train()
gen = torch.Generator(device=dev).manual_seed(0)
grid_i8 = (torch.randint(-1, 2, (2 * b, g ** 3), generator=gen, device=dev) * (torch.rand(2 * b, g ** 3, generator=gen, device=dev) < 0.3)).to(torch.int8)
small = torch.zeros(2 * b, 600 + 8192, device=dev)
rows = torch.randperm(2 * b, device=dev)[:b]
ac = encoder_ops.input_autocorr(grid_i8, g)


def step():
    f = encoder_ops.grid_encoder(small, rows, 600, g, seq, True, grid_i8=grid_i8, compact=True, autocorr=ac)
    f.backward(torch.ones_like(f))


def series(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]


for _ in range(300):
    step()
torch.cuda.synchronize()
hot = series(60)
print("hot (after 300 calls):   median %.3f ms, first 10:" % sorted(hot)[30], " ".join("%.2f" % x for x in hot[:10]))
for idle_s in (0.5, 2.0, 6.0):
    time.sleep(idle_s)
    cold = series(400)
    settle = next((i for i in range(len(cold) - 5) if max(cold[i:i + 5]) < 1.3 * sorted(hot)[30]), None)
    print("after %.1f s idle: first 24: %s" % (idle_s, " ".join("%.2f" % x for x in cold[:24])))
    print("   mean of calls 0-22 (= the old 3 + 20 protocol) %.3f ms; settles (5 calls < 1.3 x hot) at call %s = %.1f ms of GPU time; median of 400: %.3f"
          % (sum(cold[:23]) / 23, settle, sum(cold[:settle or 0]), sorted(cold)[200]))
