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


def test_evaluate_policy_and_auc_match_the_reference_fixture():
    """F12 (oracle/gen_golden_rollout.py): the reference's own evaluate_policy_grid_obs / AUC_update
    (stable_baselines3/common/evaluation.py:136-378) over a scripted 50-env 5-tuple env -- episode rewards, lengths and
    accuracies in the order the reference emits them, the mean-AUC vector, and the AUC curve step by step."""
    from tests import golden_util as gu
    from gennbv_amd.eval.evaluate import evaluate_policy_grid_obs
    fx = gu.load("F12_eval")
    n, L = int(fx["n_envs"]), int(fx["max_length"])
    rewards, dones, acc = fx["rewards"], fx["dones"], fx["accuracies"]
    calls = {"t": 0}

    class Env:
        num_envs, max_episode_length = n, L

        def reset(self):
            calls["t"] = 0
            return torch.zeros(n, 4), torch.zeros(n), torch.zeros(n), {}, {}

        def step(self, actions):
            t = calls["t"]
            calls["t"] += 1
            return (torch.full((n, 4), float(t + 1)), torch.from_numpy(rewards[t]), torch.from_numpy(dones[t]), {"episode": {}},
                    {str(i): float(acc[t, i]) for i in range(n)})

    class Model:
        class policy:  # noqa: N801
            @staticmethod
            def __call__(obs, deterministic=True):
                raise AssertionError

    m = Model()
    m.policy = lambda obs, deterministic=True: (torch.zeros(n, 6, dtype=torch.long), None, None)
    ep_r, ep_l, mean_auc, ep_acc = evaluate_policy_grid_obs(m, Env(), n_eval_episodes=n, deterministic=True, max_length=L)
    assert calls["t"] == int(fx["steps_run"])
    np.testing.assert_allclose(ep_r, fx["episode_rewards"], rtol=1e-6)
    assert list(ep_l) == list(fx["episode_lengths"])
    np.testing.assert_allclose(ep_acc, fx["episode_accuracies"], rtol=1e-6)
    np.testing.assert_allclose(mean_auc.numpy(), fx["mean_auc"], rtol=1e-6, atol=1e-7)
    # AUC_update step by step (envs that end on a step keep the column's old value; finished envs copy the previous column)
    auc, flag = torch.zeros(7, 6), torch.zeros(7)
    for t in range(6):
        d = torch.from_numpy(fx["auc2_dones"][t])
        auc = M.auc_update(auc, torch.from_numpy(fx["auc2_rewards"][t]), t + 1, d, flag)
        flag = flag + d.float()
        assert np.array_equal(auc.numpy(), fx["auc2_snapshots"][t]), t
