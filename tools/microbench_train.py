"""Time one PPO minibatch step (forward + loss + backward + clip + Adam) of the policy at G^3."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import policy_util as pu

ap = argparse.ArgumentParser()
ap.add_argument("--g", type=int, default=64)
ap.add_argument("--b", type=int, default=128)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--backend", default="torch")
ap.add_argument("--autocast", action="store_true")
ap.add_argument("--channels-last", action="store_true")
ap.add_argument("--fwd-only", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
pol, _, _ = pu.make_policy(g=a.g, device=dev, backend=a.backend, det_weights=False)
if a.channels_last:
    pol = pol.to(memory_format=torch.channels_last_3d)
d = pu.obs_dim(a.g)
obs = torch.zeros(a.b, d, device=dev)
obs[:, 600:600 + a.g ** 3] = torch.randint(-1, 2, (a.b, a.g ** 3), device=dev).float()
actions = torch.stack([torch.randint(0, n, (a.b,), device=dev) for n in pu.NVEC], -1).float()
adv = torch.randn(a.b, device=dev); ret = torch.randn(a.b, device=dev); oldv = torch.randn(a.b, device=dev); oldlp = torch.full((a.b,), -17.0, device=dev)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.autocast):
        if a.fwd_only:
            with torch.no_grad():
                pol.set_training_mode(False)
                return pol(obs)
        pol.set_training_mode(True)
        values, log_prob, entropy = pol.evaluate_actions(obs, actions)
    values = values.float().flatten()
    advn = (adv - adv.mean()) / (adv.std() + 1e-8)
    ratio = torch.exp(log_prob.float() - oldlp)
    pg = -torch.min(advn * ratio, advn * torch.clamp(ratio, 0.8, 1.2)).mean()
    vp = oldv + torch.clamp(values - oldv, -0.2, 0.2)
    loss = 10 * pg + 0.01 * (-entropy.float().mean()) + 0.8 * torch.nn.functional.mse_loss(ret, vp)
    pol.optimizer.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(pol.parameters(), 1.0)
    pol.optimizer.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.iters * 1e3
print(f"backend={a.backend} autocast={a.autocast} cl={a.channels_last} fwd_only={a.fwd_only} G={a.g} B={a.b}: {ms:.3f} ms/step")
