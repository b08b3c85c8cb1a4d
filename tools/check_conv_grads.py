"""Who is right at G=64?  torch-GPU fp32 (MIOpen) vs HIP kernels vs torch-CPU fp64 for the encoder grads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import policy_util as pu
g, b = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 4
dev = "cuda:0"
ref, _, _ = pu.make_policy(g=g, device=dev, backend="torch")
hip, _, _ = pu.make_policy(g=g, device=dev, backend="hip")
f64, _, _ = pu.make_policy(g=g, device="cpu", backend="torch")
f64 = f64.double()
gen = torch.Generator().manual_seed(g)
o = torch.zeros(b, pu.obs_dim(g))
o[:, :600] = torch.randn(b, 600, generator=gen)
o[:, 600:600 + g ** 3] = torch.randint(-1, 2, (b, g ** 3), generator=gen).float() * (torch.rand(b, g ** 3, generator=gen) < 0.4).float()
actions = torch.stack([torch.randint(0, n, (b,)) for n in pu.NVEC], -1).float()
w = torch.linspace(0.5, 1.5, b)
res = {}
for name, pol, od, dt in (("miopen", ref, dev, torch.float32), ("hip", hip, dev, torch.float32), ("f64", f64, "cpu", torch.float64)):
    pol.set_training_mode(True); pol.zero_grad()
    if dt == torch.float64:
        pol.extract_features = lambda x, p=pol: p.features_extractor(x)  # keep fp64
    v, lp, ent = pol.evaluate_actions(o.to(od, dt), actions.to(od))
    ww = w.to(od, dt)
    loss = (v.flatten() * ww).sum() + (lp * ww.flip(0)).sum() + 0.3 * (ent * ww).sum()
    loss.backward()
    res[name] = {n: p.grad.detach().double().cpu() for n, p in pol.named_parameters()}
    res[name]["values"] = v.detach().double().cpu()
for n in res["f64"]:
    r = res["f64"][n]
    s = float(r.abs().max()) + 1e-12
    print(f"{n:55s} |ref|max {s:9.3e}  miopen err {float((res['miopen'][n]-r).abs().max())/s:9.2e}  hip err {float((res['hip'][n]-r).abs().max())/s:9.2e}")
