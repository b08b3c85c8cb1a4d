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


def test_eval_callback_matches_the_reference_fixture(tmp_path):
    """F14 (oracle/gen_golden_rollout.py gen_evalcb): the reference's own EvalCallback_Grid_Obs
    (stable_baselines3/common/callbacks.py:473-708) driven for 21 on_step() calls at eval_freq = 3 over a scripted 50-env eval
    env whose rewards are rescaled per evaluation -- every logger record in order, the dump steps, the best-model saves, the
    child-callback triggers, the return values (callback_after_eval stops training at the 5th evaluation) and the contents of
    evaluations.npz, incl. the success buffer once infos carries `is_success`."""
    from tests import golden_util as gu
    from gennbv_amd.callback import BaseCallback, EvalCallback_Grid_Obs
    fx = gu.load("F14_eval_callback")
    n, L = int(fx["n_envs"]), int(fx["steps"])
    rewards, dones, acc, scales, s_from = fx["rewards"], fx["dones"], fx["accuracies"], fx["scales"], int(fx["success_from"])
    st = {"t": 0, "k": -1}

    class Env:
        num_envs, max_episode_length = n, L

        def reset(self):
            st["t"] = 0
            st["k"] += 1
            return torch.zeros(n, 4), torch.zeros(n), torch.zeros(n), {}, {}

        def step(self, actions):
            t, k = st["t"], st["k"]
            st["t"] += 1
            infos = {"episode": {}}
            if k >= s_from:
                infos["is_success"] = float((t + k) % 3 == 0)
            return (torch.full((n, 4), float(t + 1)), torch.from_numpy(rewards[t] * scales[k]), torch.from_numpy(dones[t]), infos,
                    {str(i): float(acc[t, i]) for i in range(n)})

    log = {"records": [], "dumps": [], "saves": []}

    class Logger:
        def record(self, key, value, exclude=None):
            log["records"].append((key, float(value), "" if exclude is None else str(exclude)))

        def dump(self, step=0):
            log["dumps"].append(int(step))

    class Model:
        num_timesteps = 0
        logger = Logger()
        policy = staticmethod(lambda obs, deterministic=True: (torch.zeros(n, 6, dtype=torch.long), None, None))

        def get_env(self):
            return Env()

        def save(self, path):
            import os
            log["saves"].append((os.path.basename(path), int(self.num_timesteps)))

    class Count(BaseCallback):
        def __init__(self, stop_at=None):
            super().__init__()
            self.stop_at, self.seen = stop_at, []

        def _on_step(self):
            self.seen.append(int(self.num_timesteps))
            return not (self.stop_at is not None and len(self.seen) == self.stop_at)

    on_best, after = Count(), Count(stop_at=5)
    cb = EvalCallback_Grid_Obs(Env(), callback_on_new_best=on_best, callback_after_eval=after, n_eval_episodes=n, eval_freq=int(fx["eval_freq"]),
                               log_path=str(tmp_path / "log"), best_model_save_path=str(tmp_path / "best"), verbose=0,
                               eval_kwargs=dict(max_length=L))
    st["k"] = -1
    model = Model()
    cb.init_callback(model)
    assert on_best.parent is cb and after.parent is cb and after.model is model and on_best.model is model
    rets = []
    for _ in range(int(fx["n_calls"])):
        model.num_timesteps += n
        rets.append(bool(cb.on_step()))
    assert rets == [bool(x) for x in fx["returns"]]
    assert [r[0] for r in log["records"]] == [str(k) for k in fx["record_keys"]]
    assert [r[2] for r in log["records"]] == [str(k) for k in fx["record_exclude"]]
    np.testing.assert_allclose([r[1] for r in log["records"]], fx["record_values"], rtol=2e-6)
    assert log["dumps"] == list(fx["dumps"])
    assert [s[0] for s in log["saves"]] == [str(s) for s in fx["save_names"]] and [s[1] for s in log["saves"]] == list(fx["save_timesteps"])
    assert on_best.seen == list(fx["on_best_seen"]) and after.seen == list(fx["after_seen"])
    np.testing.assert_allclose(cb.best_mean_reward, float(fx["best_mean_reward"]), rtol=2e-6)
    np.testing.assert_allclose(cb.last_mean_reward, float(fx["last_mean_reward"]), rtol=2e-6)
    z = np.load(tmp_path / "log" / "evaluations.npz")
    assert sorted(z.files) == sorted(k[4:] for k in fx.files if k.startswith("npz_"))
    assert np.array_equal(z["timesteps"], fx["npz_timesteps"]) and np.array_equal(z["ep_lengths"], fx["npz_ep_lengths"])
    np.testing.assert_allclose(z["results"], fx["npz_results"], rtol=2e-6)
    assert np.array_equal(z["successes"], fx["npz_successes"])
