"""What this chip sustains for write-only, read-only and copy streams (torch kernels over 1 GiB): the roofs the streaming kernels are priced
against in DESIGN.md (the training forward is a write stream: 252 MB of y1 + 27 MB of y2 per launch)."""
import torch
dev = "cuda:0"
n = 1 << 28  # floats: 1 GiB
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = n * 4 / 1e9
w = t(lambda: a.zero_()); print(f"write only (fill 1 GiB): {gb / w / 1e3:.2f} TB/s")
w = t(lambda: a.fill_(1.5)); print(f"write only (fill_ value): {gb / w / 1e3:.2f} TB/s")
r = t(lambda: a.sum()); print(f"read only (sum 1 GiB): {gb / r / 1e3:.2f} TB/s")
c = t(lambda: b.copy_(a)); print(f"copy (1 GiB read + 1 GiB write): {2 * gb / c / 1e3:.2f} TB/s")
c = t(lambda: torch.add(a, 1.0, out=b)); print(f"add (read + write): {2 * gb / c / 1e3:.2f} TB/s")
for mb in (64, 256):
    m = mb * (1 << 18)
    w = t(lambda: a[:m].zero_(), 50); print(f"write only, {mb} MiB: {m * 4 / 1e9 / w / 1e3:.2f} TB/s")
