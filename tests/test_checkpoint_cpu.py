"""SB3-zip checkpoints: gennbv_amd reads the archive the REFERENCE's PPO_Grid_Obs.save() wrote
(tests/golden/F10_ref_checkpoint.zip, oracle/gen_golden_ckpt.py) and round-trips its own."""
import os
import zipfile

import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests import policy_util as pu

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "F10_ref_checkpoint.zip")


def pattern(shape, dtype=torch.float32, shift=0):
    n = int(np.prod(shape)) if len(shape) else 1
    v = ((torch.arange(n) + shift) % 13 - 6).to(torch.float32) / 64.0
    return v.reshape(shape).to(dtype)


class _Env:
    """the attributes PPO_Grid_Obs reads from an env at construction"""
    num_envs, device = 4, "cpu"

    def __init__(self):
        self.observation_space, self.action_space = pu.spaces(20)

    def seed(self, s):
        pass


def make_algo(**kw):
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    args = dict(learning_rate=1e-4, n_steps=8, batch_size=8, n_epochs=3, ent_coef=0.01, vf_coef=0.8, max_grad_norm=1.0,
                target_kl=0.05, policy_kwargs=pu.policy_kwargs(20, backend="torch"), device="cpu", seed=1)
    args.update(kw)
    return PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, _Env(), **args)


def test_reads_reference_checkpoint():
    from gennbv_amd.sb3 import save_util
    data, params, variables, skipped = save_util.load_from_zip_file(FIX)
    # reference quirk: only the policy is saved (misspelt _get_th_save_params, on_policy_algorithm_grid_obs.py:299-302)
    assert set(params) == {"policy"}
    assert data["n_steps"] == 8 and data["batch_size"] == 8 and data["n_epochs"] == 3 and data["num_timesteps"] == 4096
    assert abs(data["gae_lambda"] - 0.95) < 1e-12 and data["_n_updates"] == 30
    algo = make_algo()
    algo.set_parameters(FIX, exact_match=True)
    sd = algo.policy.state_dict()
    assert list(sd.keys()) == list(params["policy"].keys())  # same names, same order as the reference's policy
    for i, (k, v) in enumerate(sd.items()):
        assert v.shape == params["policy"][k].shape and torch.equal(v, params["policy"][k]), k
        if "num_batches_tracked" in k:
            assert int(v) == 3
        elif "running_var" in k:
            assert torch.equal(v, pattern(v.shape, shift=i).abs() + 0.5), k
        else:
            # one Adam step (lr 1e-4) was applied by the reference after the pattern was set
            assert torch.allclose(v, pattern(v.shape, v.dtype, shift=i), atol=2e-4), k


def test_load_classmethod_restores_hyperparameters_and_counters():
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    algo = PPO_Grid_Obs.load(FIX, env=_Env(), device="cpu", policy_kwargs=pu.policy_kwargs(20, backend="torch"))
    assert (algo.n_steps, algo.batch_size, algo.n_epochs) == (8, 8, 3)
    assert algo.num_timesteps == 4096 and algo._n_updates == 30 and abs(algo._current_progress_remaining - 0.75) < 1e-12
    assert abs(algo.ent_coef - 0.01) < 1e-12 and abs(algo.vf_coef - 0.8) < 1e-12 and abs(algo.target_kl - 0.05) < 1e-12


def test_own_roundtrip_and_layout(tmp_path):
    a = make_algo()
    for i, p in enumerate(a.policy.parameters()):
        p.grad = pattern(p.shape, shift=i)
    a.policy.optimizer.step()
    a.num_timesteps, a._n_updates = 123, 7
    path = str(tmp_path / "ckpt")
    a.save(path)  # default: the reference's layout (policy only)
    with zipfile.ZipFile(path + ".zip") as z, zipfile.ZipFile(FIX) as zr:
        assert set(z.namelist()) == set(zr.namelist()) == {"data", "pytorch_variables.pth", "policy.pth",
                                                           "_stable_baselines3_version", "system_info.txt"}
    full = str(tmp_path / "ckpt_full")
    a.save(full, include_optimizer=True)
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    with pytest.raises(ValueError):  # policy_kwargs names a class outside the allow-list (tests.torch_reference)
        PPO_Grid_Obs.load(full, env=_Env(), device="cpu")
    b = PPO_Grid_Obs.load(full, env=_Env(), device="cpu", trusted=True)
    assert b.num_timesteps == 123 and b._n_updates == 7 and b.n_steps == a.n_steps
    for (k, x), (_, y) in zip(a.policy.state_dict().items(), b.policy.state_dict().items()):
        assert torch.equal(x, y), k
    sa, sb = a.policy.optimizer.state_dict()["state"], b.policy.optimizer.state_dict()["state"]
    assert len(sa) == len(sb) > 0
    for i in sa:
        assert torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) and torch.equal(sa[i]["exp_avg_sq"], sb[i]["exp_avg_sq"])


def test_best_checkpoint_callback_protocol(tmp_path):
    """gennbv/callback.py:25-70: periodic checkpoint at rollout end + `<prefix>_best_<key>` on a new maximum of the
    mean episode-info value; driven through the hooks learn()/collect_rollouts() call."""
    from collections import deque
    from gennbv_amd.callback import BestCKPTCallback
    saved = []

    class _Model:
        num_timesteps = 0
        ep_info_buffer = deque(maxlen=100)
        logger = type("L", (), {})()

        def save(self, path):
            saved.append(os.path.basename(path))

    m = _Model()
    cb = BestCKPTCallback(save_freq=4, save_path=str(tmp_path), name_prefix="gennbv", key_list=["episode_reward"])
    cb.on_training_start({"self": m}, {})
    for rollout, rew in enumerate([1.0, 3.0, 2.0]):
        cb.on_rollout_start()
        for _ in range(2):
            m.num_timesteps += 8
            cb.update_locals({})
            assert cb.on_step() is True
        m.ep_info_buffer.append({"episode_reward": torch.tensor(rew), "episode_length": 4.0})
        cb.on_rollout_end()
    cb.on_training_end()
    # rollout 0: mean 1 -> best; rollout 1: n_calls = 4 -> periodic, mean 2 -> best; rollout 2: mean 2 -> no new best
    assert saved == ["gennbv_best_episode_reward", "gennbv_32_steps", "gennbv_best_episode_reward"]
    assert abs(cb.key_highest_value["episode_reward"] - 2.0) < 1e-6


def test_untrusted_archive_cannot_execute_code(tmp_path):
    """A ':serialized:' blob whose pickle calls os.system must be skipped by the restricted unpickler."""
    import base64, json, pickle
    from gennbv_amd.sb3 import save_util
    marker = tmp_path / "pwned"

    class _Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))
    text = json.dumps({"n_steps": 8, "policy_kwargs": {":type:": "x", ":serialized:": base64.b64encode(pickle.dumps(_Evil())).decode()},
                       "plain": {":type:": "dict", ":serialized:": base64.b64encode(pickle.dumps({"a": [1, 2.5, "s"]})).decode()}})
    data, skipped = save_util.json_to_data(text)
    assert skipped == ["policy_kwargs"] and data["plain"] == {"a": [1, 2.5, "s"]} and data["n_steps"] == 8
    assert not marker.exists()


def _proto4_global_call(module: str, name: str, arg: str) -> bytes:
    """Hand-built protocol-4 pickle: STACK_GLOBAL(module, name)(arg)."""
    def s(x):
        b = x.encode()
        return b"\x8c" + bytes([len(b)]) + b
    return b"\x80\x04" + s(module) + s(name) + b"\x93" + s(arg) + b"\x85R."


@pytest.mark.parametrize("module,name", [("gennbv_amd.sb3.save_util", "os.system"), ("gennbv_amd.sb3.save_util", "_loads"),
                                         ("gennbv_amd.sb3.save_util", "pickle.loads"), ("gennbv_amd._lib", "ctypes.CDLL"),
                                         ("gennbv_amd.spaces", "np.load"), ("gennbv_amd.sb3.save_util", "os")])
def test_untrusted_archive_dotted_and_package_gadgets_are_rejected(tmp_path, module, name):
    """The allow-list is exact (module, name) pairs: a package-prefix rule let protocol-4 dotted names reach os.system
    through any module of the package."""
    import pickle
    from gennbv_amd.sb3 import save_util
    marker = tmp_path / "pwned"
    blob = _proto4_global_call(module, name, f"touch {marker}")
    with pytest.raises(pickle.UnpicklingError):
        save_util._loads(blob, False)
    assert not marker.exists()


def test_untrusted_archive_keeps_the_package_data_classes():
    import pickle
    import numpy as np
    from gennbv_amd.sb3 import save_util
    from gennbv_amd.spaces import Box, MultiDiscrete
    from gennbv_amd.network.hybrid_encoder import Hybrid_Encoder
    obj = {"observation_space": Box(-np.inf, np.inf, (5,)), "action_space": MultiDiscrete([3, 4]),
           "policy_kwargs": {"features_extractor_class": Hybrid_Encoder, "net_arch": [256]}}
    out = save_util._loads(pickle.dumps(obj, protocol=4), False)
    assert out["observation_space"].shape == (5,) and out["action_space"].nvec.tolist() == [3, 4]
    assert out["policy_kwargs"]["features_extractor_class"] is Hybrid_Encoder
