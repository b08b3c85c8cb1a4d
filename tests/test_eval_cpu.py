"""Host side of the evaluation path: AUC bookkeeping and the 1-cm point de-duplication against
restatements of the reference (oracle.auc_update_ref = evaluation.py:358-378 env by env)."""
import numpy as np
import torch

from gennbv_amd.eval import metrics as M
from oracle import oracle


def test_auc_update_equals_the_reference_loop():
    rng = np.random.default_rng(0)
    n, L = 7, 6
    auc = torch.zeros(n, L)
    ref = np.zeros((n, L), np.float32)
    flag = np.zeros(n)
    for step in range(1, L + 1):
        rew = rng.random(n).astype(np.float32)
        dones = (rng.random(n) < 0.3).astype(np.int64) * (flag == 0)
        ref = oracle.auc_update_ref(ref, rew, step, dones, flag)
        auc = M.auc_update(auc, torch.from_numpy(rew), step, torch.from_numpy(dones), torch.from_numpy(flag))
        flag = flag + dones
        assert np.array_equal(auc.numpy(), ref), step
    w = np.array([L - i for i in range(L)], np.float32)
    assert np.allclose(M.mean_auc(auc).numpy(), (ref * w).sum(1) / L, rtol=1e-6)


def test_unique_rounded_points_equals_torch_unique_of_round():
    g = torch.Generator().manual_seed(1)
    pts = (torch.rand(5000, 3, generator=g) - 0.5) * 3.0
    pts = torch.cat([pts, pts[:700] + 0.003, -pts[:5]])  # near-duplicates collapse at 1 cm
    got = M.unique_rounded_points(pts, 2)
    ref = torch.unique(torch.round(pts, decimals=2), dim=0)
    assert got.shape == ref.shape and torch.equal(got, ref)
    assert M.unique_rounded_points(torch.zeros(0, 3)).shape == (0, 3)
    far = torch.tensor([[2.0e5, 0.0, 0.0], [2.0e5, 0.0, 0.0]])  # outside the 21-bit key range: library path
    assert M.unique_rounded_points(far).shape == (1, 3)
