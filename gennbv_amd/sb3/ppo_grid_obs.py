"""PPO_Grid_Obs: the GenNBV PPO (stable_baselines3/ppo/ppo_grid_obs.py:75-322) with its
on-policy loop (stable_baselines3/common/on_policy_algorithm_grid_obs.py:102-298) and the
pieces of the algorithm base it needs (base_class_grid_obs.py:408-477, :600-614).

Same constructor signature, `.train()`, `.learn()`, `.collect_rollouts()` contracts and the
exact loss of the reference:

    adv   = (A - mean) / (std_unbiased + 1e-8)            per minibatch          (:214-216)
    ratio = exp(logp - logp_old); pg = -mean(min(adv*ratio, adv*clamp(ratio, 1-c, 1+c)))
    v_pred = v_old + clamp(v - v_old, -c_vf, c_vf); vl = mse(returns, v_pred)  (:231-241)
    loss  = 10*pg + ent_coef*(-mean(entropy)) + vf_coef*vl                       (:253)
    approx_kl = mean(exp(lr) - 1 - lr); stop when > 1.5*target_kl (before the step) (:259-268)
    clip_grad_norm_(max_grad_norm); Adam(eps=1e-5)                                (:271-275)

MI355X-first differences (all observationally equivalent, DESIGN.md section "PPO"):
  * collect_rollouts evaluates the policy ONCE per env step: the reference computes
    V(new_obs) for the time-out bootstrap (:205-208) and then, at the next step, runs the
    full policy on the very same observation with the very same (eval-mode) parameters
    (:168); both values come from one forward here. RNG consumption order is unchanged.
  * the env writes each observation straight into the rollout buffer row it will be
    stored in (`step(..., obs_out=...)`), the GAE scan is one kernel, the buffer is
    allocated once;
  * logging scalars are accumulated on the device and read back once per train() call
    instead of five host syncs per minibatch; the KL early-stop decision is the only
    per-minibatch read (and can be polled per epoch with `kl_poll="epoch"`, which masks
    the updates after the stopping minibatch on the device).
"""
from __future__ import annotations

import time
from typing import Any, Dict, Optional, Union

import os

import numpy as np
import torch
import torch.nn.functional as F

from .buffers import TensorRolloutBuffer_Grid_Obs
from .logger import Logger
from ._train_hip import FusedTrainMixin
from ._dp_step import DataParallelStepMixin
from ._capture import GraphCaptureMixin
from ._rollout import RolloutMixin


def _schedule(v):
    if isinstance(v, (float, int)):
        val = float(v)
        return lambda _: val
    assert callable(v)
    return v


class PPO_Grid_Obs(RolloutMixin, FusedTrainMixin, DataParallelStepMixin, GraphCaptureMixin):
    def __init__(self, policy, env, learning_rate=3e-4, n_steps: int = 2048, batch_size: int = 64, n_epochs: int = 10,
                 gamma: float = 0.99, gae_lambda: float = 0.95, clip_range=0.2, clip_range_vf=None,
                 normalize_advantage: bool = True, ent_coef: float = 0.0, vf_coef: float = 0.5, max_grad_norm: float = 0.5,
                 use_sde: bool = False, sde_sample_freq: int = -1, target_kl: Optional[float] = None,
                 tensorboard_log: Optional[str] = None, create_eval_env: bool = False,
                 policy_kwargs: Optional[Dict[str, Any]] = None, verbose: int = 0, seed: Optional[int] = None,
                 device: Union[torch.device, str] = "auto", _init_setup_model: bool = True, compact_obs: Optional[bool] = None):
        """`compact_obs` (additive, default off): keep the tri-class grid of every stored
        observation as int8 only (sb3/buffers.py `compact`) -- same values, 3.6x less HBM for the rollout buffer."""
        assert not use_sde, "gSDE is not on the GenNBV path"
        if normalize_advantage:
            assert batch_size > 1, "`batch_size` must be greater than 1. See https://github.com/DLR-RM/stable-baselines3/issues/440"
        self.policy_class, self.env = policy, env
        self.policy_kwargs = {} if policy_kwargs is None else policy_kwargs
        self.learning_rate, self.n_steps, self.batch_size, self.n_epochs = learning_rate, n_steps, batch_size, n_epochs
        self.gamma, self.gae_lambda = gamma, gae_lambda
        self.clip_range, self.clip_range_vf = clip_range, clip_range_vf
        self.normalize_advantage, self.ent_coef, self.vf_coef = normalize_advantage, ent_coef, vf_coef
        self.max_grad_norm, self.target_kl = max_grad_norm, target_kl
        self.verbose, self.seed = verbose, seed
        if device == "auto":
            device = getattr(env, "device", "cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(device)
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.n_envs = env.num_envs
        if self.env is not None:
            buffer_size = self.n_envs * self.n_steps
            assert buffer_size > 1
        self.num_timesteps = 0
        self._n_updates = 0
        self._current_progress_remaining = 1.0
        self._last_obs = None
        self._last_episode_starts = None
        self._pending = None  # (actions, values, log_probs) already evaluated on _last_obs
        self.ep_info_buffer = None
        self._logger = Logger(verbose)
        self.policy_loss_scale = 10.0  # ppo_grid_obs.py:253
        self.kl_poll = "minibatch"
        self.train_impl = "hip"   # fused loss / flat Adam / device-side early stop when the encoder backend is "hip"
        self.use_graph = True     # replay the minibatch step as one hipGraph
        self.graph_candidates = None  # None: 3 when a train() call replays the graph >= 256 times, else 1 (see _capture_minibatch_graph)
        self.grad_write_through = True  # backward kernels store into the flat gradient buffer (ops/direct_grad.py)
        self.rotate_rows = True   # replayed graph on one GPU: the Adam launch leaves the next minibatch's row numbers behind (no host copy)
        self.grid_i8_rows = True  # int8 side copy of the grid rows next to flat fp32 rows (what the conv1 kernels of the update read)
        # (the data-parallel step has ONE order since round 6: phase A's second-stream work is joined by the gradient exchange, not by the conv
        # backward, and the rotation table is used inside the RCCL graph as well; the round-4 order behind GENNBV_DP_LATE_ASIDE / GENNBV_DP_ROTATE
        # lost both of its measurements -- 847-869 against 804-816 ms per iteration, profiles/r05_notes.md section 9 -- and is gone)
        # seconds between the eager warm-up collectives and the first RCCL capture (GENNBV_DP_SETTLE; see _capture_minibatch_graph)
        self.dp_capture_settle_s = float(os.environ.get("GENNBV_DP_SETTLE", "0.35"))
        self.dp_stress_spin_cycles = 0  # > 0: tests only, see _dp_minibatch
        self.fused_add = True     # time-out bootstrap + the five buffer copies of rollout_buffer.add as one launch (gnbv_rollout_add)
        self.rollout_plan = True  # collect_rollouts evaluates the policy through ops/rollout_plan.py (same kernels, step-invariant work hoisted)
        self.compact_obs = bool(compact_obs)
        # Time-out bootstrap (on_policy_algorithm_grid_obs.py:205-208).  "reference": what the reference computes --
        # `predict_values(new_obs)[0]` is ROW 0 of the [N, 1] values, so every timed-out env is bootstrapped with env 0's
        # value (pinned by fixture F11).  "per_env": each env's own V(new_obs), SB3's evident intent (not the reference).
        self.timeout_bootstrap = "reference"
        self._hip = None
        self._sync = None         # gennbv_amd.parallel.GradSync when data-parallel
        if _init_setup_model:
            self._setup_model()

    # ------------------------------------------------------------------------------
    @property
    def logger(self):
        return self._logger

    def set_random_seed(self, seed: Optional[int] = None) -> None:
        """base_class_grid_obs.py:600-614 / common/utils.py:25-42."""
        if seed is None:
            return
        import random
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.action_space_seed = seed
        if self.env is not None:
            self.env.seed(seed)

    def _setup_model(self) -> None:
        self.lr_schedule = _schedule(self.learning_rate)
        self.set_random_seed(self.seed)
        # (the reference builds the buffer first, :base 408-477; neither step draws from an RNG the other uses)
        self.policy = self.policy_class(self.observation_space, self.action_space, self.lr_schedule, use_sde=False,
                                        **self.policy_kwargs).to(self.device)
        enc = self.policy.features_extractor
        hip = getattr(enc, "backend", "torch") == "hip"
        compact = None
        if self.compact_obs:
            if not (getattr(self.env, "supports_grid_i8", False) and hip and self.device.type == "cuda"):
                raise ValueError("compact_obs needs an env that writes the int8 grid rows (supports_grid_i8), the gfx950 "
                                 "encoder and a GPU device")
            compact = (int(enc.state_input_shape[0]), int(enc.grid_size) ** 3)
        self.rollout_buffer = TensorRolloutBuffer_Grid_Obs(self.n_steps, self.observation_space, self.action_space,
                                                           device=self.device, gamma=self.gamma,
                                                           gae_lambda=self.gae_lambda, n_envs=self.n_envs, compact=compact)
        self.rollout_buffer.lazy_obs = hip
        self.clip_range = _schedule(self.clip_range)
        if self.clip_range_vf is not None:
            if isinstance(self.clip_range_vf, (float, int)):
                assert self.clip_range_vf > 0, "`clip_range_vf` must be positive, pass `None` to deactivate vf clipping"
            self.clip_range_vf = _schedule(self.clip_range_vf)

    # ------------------------------------------------------------------------------
    # SB3-zip checkpoints (base_class_grid_obs.py:616-854, on_policy_algorithm_grid_obs.py:300-303)
    def _get_torch_save_params(self):
        # Reference quirk kept: its on-policy class overrides `_get_th_save_params` (a misspelt name,
        # on_policy_algorithm_grid_obs.py:299-302), so the base list ["policy"] is what save()/set_parameters()
        # use -- the reference's checkpoints carry NO optimizer state and its exact-match check rejects archives
        # that do.  save(include_optimizer=True) adds policy.optimizer.pth for this build's own resume.
        return ["policy"], []

    def get_parameters(self) -> Dict[str, Dict]:
        """{"policy": state_dict, "policy.optimizer": torch.optim.Adam-format state dict}."""
        opt_sd = self.policy.optimizer.state_dict()
        if self._hip and self._hip.get("opt") is not None:
            # (NOT a collective: with the sharded fc_grid update every rank's moments are made complete at the end of each train() call
            # -- `_train_hip_once` -- so a save on one rank only, or on rank-local conditions inside callbacks, cannot hang the others)
            opt_sd = self._hip["opt"].torch_state_dict(self.policy.optimizer)  # the flat HIP Adam owns the moments
        return {"policy": self.policy.state_dict(), "policy.optimizer": opt_sd}

    def set_parameters(self, load_path_or_dict, exact_match: bool = True, device="auto") -> None:
        """In-place load of a zip written by this class OR by the reference's PPO_Grid_Obs.save()."""
        from . import save_util
        params = load_path_or_dict
        if not isinstance(load_path_or_dict, dict):
            _, params, _, _ = save_util.load_from_zip_file(load_path_or_dict, load_data=False,
                                                           device=self.device if device == "auto" else device)
        updated = set()
        for name, sd in params.items():
            if name == "policy":
                self.policy.load_state_dict(sd, strict=exact_match)
            elif name == "policy.optimizer":
                self.policy.optimizer.load_state_dict(sd)
                if self._hip and self._hip.get("opt") is not None:
                    self._hip["opt"].load_torch_state_dict(sd)
                    self._hip["graph"] = None
            else:
                raise ValueError(f"Key {name} is an invalid object name.")
            updated.add(name)
        if exact_match and not ({"policy"} <= updated <= {"policy", "policy.optimizer"}):
            raise ValueError(f"Names of parameters do not match agents' parameters: expected "
                             f"{set(self._get_torch_save_params()[0])} (+ optionally 'policy.optimizer'), got {updated}")

    def save(self, path, exclude=None, include=None, include_optimizer: bool = False) -> None:
        """base_class_grid_obs.py:806-854: `data` JSON + policy.pth (+ policy.optimizer.pth on request) in one zip."""
        from . import save_util
        excl = set(exclude or []) | save_util.EXCLUDED
        if include is not None:
            excl -= set(include)
        data = {k: v for k, v in self.__dict__.items() if k not in excl}
        params = {k: ({kk: vv.detach().cpu() for kk, vv in v.items()} if k == "policy" else v) for k, v in self.get_parameters().items()
                  if k == "policy" or include_optimizer}
        save_util.save_to_zip_file(path, data=data, params=params, pytorch_variables={})

    @classmethod
    def load(cls, path, env=None, device="auto", custom_objects=None, policy=None, trusted: bool = False, **kwargs) -> "PPO_Grid_Obs":
        """Re-create the algorithm from a zip.  Plain hyper-parameters come from `data`; entries pickled with
        classes that are not importable here (the reference's policy class, spaces, schedules) are replaced by
        `policy` / the env's spaces / kwargs.  Pickled entries of `data` pass a restricted unpickler unless
        `trusted=True` (sb3/save_util.py)."""
        from . import save_util
        data, params, _, skipped = save_util.load_from_zip_file(path, custom_objects=custom_objects, device="cpu", trusted=trusted)
        data = data or {}
        if env is None:
            raise ValueError("PPO_Grid_Obs.load needs the (replay-feed) env: stored environments are not restored")
        policy_class = policy or data.get("policy_class")
        if policy_class is None or isinstance(policy_class, dict):
            from .policies import ActorCriticPolicy_Train_Eval
            policy_class = ActorCriticPolicy_Train_Eval
        ctor = {}
        for k in ("learning_rate", "n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "clip_range", "clip_range_vf",
                  "normalize_advantage", "ent_coef", "vf_coef", "max_grad_norm", "target_kl", "policy_kwargs", "verbose", "seed"):
            if k in data and k not in skipped:
                ctor[k] = data[k]
        ctor.update(kwargs)
        if "policy_kwargs" not in ctor:
            raise ValueError("the archive's policy_kwargs could not be restored (entries skipped by the restricted unpickler: "
                             f"{skipped}); pass policy_kwargs=... or, for an archive you trust, trusted=True")
        model = cls(policy_class, env, device=device, **ctor)
        for k in ("num_timesteps", "_n_updates", "_current_progress_remaining"):
            if k in data:
                setattr(model, k, data[k])
        model.set_parameters(params, exact_match=True)
        return model

    def _update_learning_rate(self, optimizer) -> None:
        lr = self.lr_schedule(self._current_progress_remaining)
        self.logger.record("train/learning_rate", lr)
        for group in optimizer.param_groups:
            group["lr"] = lr

    # ------------------------------------------------------------------------------
    def train(self) -> None:
        """Update the policy on the gathered rollout buffer (ppo_grid_obs.py:176-297)."""
        if getattr(self.policy.features_extractor, "backend", "torch") == "hip" and self.train_impl == "hip":
            return self._train_hip()
        training_start = time.time()
        self.policy.set_training_mode(True)
        self._update_learning_rate(self.policy.optimizer)
        clip_range = self.clip_range(self._current_progress_remaining)
        clip_range_vf = None if self.clip_range_vf is None else self.clip_range_vf(self._current_progress_remaining)
        stats = []  # per minibatch: [pg, vl, ent, kl, clip_fraction, loss] on the device
        kl_per_epoch = []
        continue_training = True
        for epoch in range(self.n_epochs):
            epoch_kl = []
            for rollout_data in self.rollout_buffer.get(self.batch_size):
                actions = rollout_data.actions
                values, log_prob, entropy = self.policy.evaluate_actions(rollout_data.observations, actions)
                values = values.flatten()
                advantages = rollout_data.advantages
                if self.normalize_advantage and self._sync is not None and self._sync.active:
                    # data-parallel: the statistics of the GLOBAL minibatch (all ranks' rows), gennbv_amd/parallel.py
                    gm, ginv = self._sync.global_adv_norm(advantages.view(1, -1))[0]
                    advantages = (advantages - gm) * ginv
                elif self.normalize_advantage:
                    advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
                ratio = torch.exp(log_prob - rollout_data.old_log_prob)
                policy_loss_1 = advantages * ratio
                policy_loss_2 = advantages * torch.clamp(ratio, 1 - clip_range, 1 + clip_range)
                policy_loss = -torch.min(policy_loss_1, policy_loss_2).mean()
                clip_fraction = torch.mean((torch.abs(ratio - 1) > clip_range).float())
                if clip_range_vf is None:
                    values_pred = values
                else:
                    values_pred = rollout_data.old_values + torch.clamp(values - rollout_data.old_values, -clip_range_vf,
                                                                        clip_range_vf)
                value_loss = F.mse_loss(rollout_data.returns, values_pred)
                entropy_loss = -torch.mean(entropy)
                loss = policy_loss * self.policy_loss_scale + self.ent_coef * entropy_loss + self.vf_coef * value_loss
                with torch.no_grad():
                    log_ratio = log_prob - rollout_data.old_log_prob
                    approx_kl_div = torch.mean((torch.exp(log_ratio) - 1) - log_ratio)
                    if self._sync is not None:  # KL of the global minibatch: every rank stops together
                        approx_kl_div = self._sync.mean_scalar(approx_kl_div)
                stats.append(torch.stack([policy_loss.detach(), value_loss.detach(), entropy_loss.detach(), approx_kl_div,
                                          clip_fraction, loss.detach()]))
                epoch_kl.append(len(stats) - 1)
                if self.target_kl is not None and float(approx_kl_div) > 1.5 * self.target_kl:
                    continue_training = False
                    if self.verbose >= 1:
                        print(f"Early stopping at step {epoch} due to reaching max kl: {float(approx_kl_div):.2f}")
                    break
                self.policy.optimizer.zero_grad()
                loss.backward()
                if self._sync is not None:
                    self._sync.average_grads(self.policy.parameters())
                torch.nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
                self.policy.optimizer.step()
            kl_per_epoch.append(epoch_kl)
            if not continue_training:
                break
        self._n_updates += self.n_epochs
        s = torch.stack(stats).double().cpu().numpy()  # ONE read-back for all logging scalars
        self.last_train_stats = s
        v_flat, r_flat = self.rollout_buffer.flat_values_returns()
        var_y = torch.var(r_flat, unbiased=False)
        explained_var = float("nan") if float(var_y) == 0 else float(1 - torch.var(r_flat - v_flat, unbiased=False) / var_y)
        self.logger.record("train/entropy_loss", float(np.mean(s[:, 2])))
        self.logger.record("train/policy_gradient_loss", float(np.mean(s[:, 0])))
        self.logger.record("train/value_loss", float(np.mean(s[:, 1])))
        self.logger.record("train/approx_kl", float(np.mean(s[kl_per_epoch[-1], 3])))  # last epoch's list (:241)
        self.logger.record("train/clip_fraction", float(np.mean(s[:, 4])))
        self.logger.record("train/loss", float(s[-1, 5]))
        self.logger.record("train/explained_variance", explained_var)
        self.logger.record("train/n_updates", self._n_updates)
        self.logger.record("train/clip_range", clip_range)
        if clip_range_vf is not None:
            self.logger.record("train/clip_range_vf", clip_range_vf)
        self.logger.record("time/training", time.time() - training_start)

    # ------------------------------------------------------------------------------


    # ------------------------------------------------------------------------------


    def _setup_learn(self, total_timesteps: int, reset_num_timesteps: bool = True):
        """base_class_grid_obs.py:408-477."""
        from collections import deque
        self.start_time = time.time()
        if self.ep_info_buffer is None or reset_num_timesteps:
            self.ep_info_buffer = deque(maxlen=100)
        if reset_num_timesteps:
            self.num_timesteps = 0
        else:
            total_timesteps += self.num_timesteps
        self._total_timesteps = total_timesteps
        self._maybe_enable_grid_i8()
        if reset_num_timesteps or self._last_obs is None:
            try:
                if self.rollout_buffer.grid_i8 is not None:
                    self._last_obs = self.env.reset(obs_out=self.rollout_buffer.first_obs_row(), grid_i8_out=self.rollout_buffer.grid_i8[0])
                    self.rollout_buffer.update_autocorr(0)
                else:
                    self._last_obs = self.env.reset(obs_out=self.rollout_buffer.first_obs_row())
            except TypeError:
                self._last_obs = self.env.reset()
            self._last_episode_starts = torch.ones(self.env.num_envs, dtype=torch.bool, device=self.device)
            self._pending = None
        # "keep same with PPO from RSL-rl" (:470-475): de-synchronise the episodes
        elb = self.env.episode_length_buf
        self.env.episode_length_buf = torch.randint_like(elb, high=int(self.env.max_episode_length))
        return total_timesteps

    def get_env(self):
        return self.env

    def get_vec_normalize_env(self):
        return None  # tensor envs are never wrapped in VecNormalize (base_class_grid_obs.py:511-518)

    def _init_callback(self, callback, eval_env=None, eval_freq: int = 10000, n_eval_episodes: int = 5, log_path: Optional[str] = None):
        """base_class_grid_obs.py:370-406: a list becomes a CallbackList, a function a ConvertCallback, an `eval_env`
        adds an EvalCallback_Grid_Obs; the callback tree is bound to this model."""
        from ..callback import BaseCallback, CallbackList, ConvertCallback, EvalCallback_Grid_Obs
        if callback is None and eval_env is None:
            return None
        if isinstance(callback, list):
            callback = CallbackList(callback)
        if not isinstance(callback, BaseCallback):
            callback = ConvertCallback(callback)
        if eval_env is not None:
            callback = CallbackList([callback, EvalCallback_Grid_Obs(eval_env, best_model_save_path=log_path, log_path=log_path,
                                                                      eval_freq=eval_freq, n_eval_episodes=n_eval_episodes)])
        callback.init_callback(self)
        return callback

    def _update_current_progress_remaining(self, num_timesteps: int, total_timesteps: int) -> None:
        self._current_progress_remaining = 1.0 - float(num_timesteps) / float(total_timesteps)

    def learn(self, total_timesteps: int, callback=None, log_interval: int = 1, eval_env=None, eval_freq: int = -1,
              n_eval_episodes: int = 5, tb_log_name: str = "PPO", eval_log_path: Optional[str] = None,
              reset_num_timesteps: bool = True) -> "PPO_Grid_Obs":
        """on_policy_algorithm_grid_obs.py:230-298."""
        iteration = 0
        total_timesteps = self._setup_learn(total_timesteps, reset_num_timesteps)
        callback = self._init_callback(callback, eval_env, eval_freq, n_eval_episodes, eval_log_path)
        if callback is not None:
            callback.on_training_start(locals(), globals())
        while self.num_timesteps < total_timesteps:
            collect_start_time = time.time()
            continue_training = self.collect_rollouts(self.env, callback, self.rollout_buffer, n_rollout_steps=self.n_steps)
            if continue_training is False:
                break
            iteration += 1
            self._update_current_progress_remaining(self.num_timesteps, total_timesteps)
            if log_interval is not None and iteration % log_interval == 0:
                if self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)
                time_to_collect = time.time() - collect_start_time
                fps = int(self.rollout_buffer.buffer_size * self.rollout_buffer.n_envs / max(time_to_collect, 1e-9))
                self.logger.record("time/iterations", iteration)
                if len(self.ep_info_buffer) > 0 and self.ep_info_buffer[-1] is not None:
                    for key, val in dict(self.ep_info_buffer[-1].items()).items():
                        self.logger.record("rollout/{}".format(key), float(val))
                self.logger.record("time/fps", fps)
                self.logger.record("time/time_elapsed", int(time.time() - self.start_time))
                self.logger.record("time/total_timesteps", self.num_timesteps)
                self.logger.record("time/rollout", time_to_collect)
                self.logger.dump(step=self.num_timesteps)
            self.train()
        if callback is not None:
            callback.on_training_end()
        return self
