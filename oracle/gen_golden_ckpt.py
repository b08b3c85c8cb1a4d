"""TEST INFRASTRUCTURE (this container only): golden SB3-zip checkpoint written by the REFERENCE's own
`PPO_Grid_Obs.save()` (base_class_grid_obs.py:806-854), for tests/test_checkpoint_cpu.py.

    python oracle/gen_golden_ckpt.py     ->  tests/golden/F10_ref_checkpoint.zip  (+ cross-load check)

The policy weights are a low-entropy deterministic pattern ((i mod 13) - 6) / 64 per tensor (one Adam step
on a patterned gradient applied), so that the archive deflates to a few tens of KB.  NOTE the reference
saves ONLY policy.pth: its on-policy class overrides a misspelt `_get_th_save_params` (the reference writes ZIP_STORED; the fixture is the same archive re-compressed member by member).
Also verifies the other direction here, where the reference is importable: the reference's
`set_parameters()` reads a zip written by gennbv_amd's `PPO_Grid_Obs.save()`.
"""
import io
import os
import sys
import tempfile
import types
import zipfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
import gen_golden_ppo as gp  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pattern(shape, dtype=torch.float32, shift=0):
    n = int(np.prod(shape)) if len(shape) else 1
    v = ((torch.arange(n) + shift) % 13 - 6).to(torch.float32) / 64.0
    return v.reshape(shape).to(dtype)


def patterned_policy(ref):
    pol, obs_space, act_space = gp.make_policy(ref, seed=0)
    sd = pol.state_dict()
    new = {}
    for i, (k, v) in enumerate(sd.items()):
        if "num_batches_tracked" in k:
            new[k] = torch.tensor(3, dtype=v.dtype)
        elif "running_var" in k:
            new[k] = pattern(v.shape, shift=i).abs() + 0.5
        else:
            new[k] = pattern(v.shape, v.dtype, shift=i)
    pol.load_state_dict(new)
    pol.optimizer = torch.optim.Adam(pol.parameters(), lr=1e-4, eps=1e-5)
    for i, p in enumerate(pol.parameters()):
        p.grad = pattern(p.shape, shift=3 * i + 1)
    pol.optimizer.step()
    return pol, obs_space, act_space


def ref_ppo(ref, pol, obs_space, act_space):
    PPO = ref.ppo_grid_obs.PPO_Grid_Obs
    ppo = object.__new__(PPO)
    ppo.policy = pol
    ppo.policy_class = type(pol)
    ppo.policy_kwargs = {"net_arch": []}
    ppo.observation_space, ppo.action_space = obs_space, act_space
    ppo.n_envs, ppo.n_steps, ppo.batch_size, ppo.n_epochs = 4, 8, 8, 3
    ppo.gamma, ppo.gae_lambda, ppo.ent_coef, ppo.vf_coef = 0.99, 0.95, 0.01, 0.8
    ppo.max_grad_norm, ppo.target_kl, ppo.normalize_advantage = 1.0, 0.05, True
    ppo.learning_rate = 1e-4
    ppo.lr_schedule = lambda _: 1e-4
    ppo.clip_range = lambda _: 0.2
    ppo.clip_range_vf = None
    ppo.use_sde, ppo.sde_sample_freq, ppo.verbose, ppo.seed = False, -1, 0, 1
    ppo.num_timesteps, ppo._n_updates, ppo._current_progress_remaining = 4096, 30, 0.75
    ppo.device = torch.device("cpu")
    ppo.env = None
    return ppo, PPO


def recompress(src_bytes: bytes, dst: str):
    with zipfile.ZipFile(io.BytesIO(src_bytes)) as zi, zipfile.ZipFile(dst, "w", zipfile.ZIP_DEFLATED, compresslevel=9) as zo:
        for info in zi.infolist():
            zo.writestr(info.filename, zi.read(info.filename))


if __name__ == "__main__":
    ref = ref_harness.import_reference()
    pol, obs_space, act_space = patterned_policy(ref)
    ppo, PPO = ref_ppo(ref, pol, obs_space, act_space)
    buf = io.BytesIO()
    PPO.save(ppo, buf)
    os.makedirs(GOLDEN, exist_ok=True)
    dst = os.path.join(GOLDEN, "F10_ref_checkpoint.zip")
    recompress(buf.getvalue(), dst)
    print("reference zip", len(buf.getvalue()), "bytes ->", os.path.getsize(dst), "bytes deflated")
    with zipfile.ZipFile(dst) as z:
        print("members:", z.namelist())

    # ---- other direction: the reference reads a zip written by gennbv_amd ----
    from tests import policy_util as pu
    from tests.test_checkpoint_cpu import make_algo
    ours = make_algo()
    with torch.no_grad():
        for i, p in enumerate(ours.policy.parameters()):
            p.copy_(pattern(p.shape, shift=5 * i + 2))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ours.zip")
        ours.save(path)
        pol2, _, _ = gp.make_policy(ref, seed=1)
        pol2.optimizer = torch.optim.Adam(pol2.parameters(), lr=1e-4, eps=1e-5)
        ppo2, _ = ref_ppo(ref, pol2, obs_space, act_space)
        PPO.set_parameters(ppo2, path, exact_match=True, device="cpu")
        for (k, a), (_, b) in zip(ours.policy.state_dict().items(), pol2.state_dict().items()):
            assert torch.equal(a.cpu(), b), k
    print("cross-load OK: reference set_parameters() read a gennbv_amd checkpoint bit-exactly")
