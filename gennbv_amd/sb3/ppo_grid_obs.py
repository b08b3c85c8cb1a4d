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


def _schedule(v):
    if isinstance(v, (float, int)):
        val = float(v)
        return lambda _: val
    assert callable(v)
    return v


class PPO_Grid_Obs:
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
    def _hip_setup(self, batch: int, n_minibatches: int):
        from ..ops.ppo_ops import FlatAdam, PpoLossOp
        pc = self
        loss = PpoLossOp(batch, list(self.action_space.nvec), self.device, self.n_epochs * n_minibatches,
                         self.clip_range(1.0), None if self.clip_range_vf is None else self.clip_range_vf(1.0),
                         self.ent_coef, self.vf_coef, self.policy_loss_scale, self.target_kl, self.normalize_advantage)
        opt = self._hip["opt"] if self._hip else None
        if opt is None:
            old = self.policy.optimizer
            opt = FlatAdam(self.policy, lr=self.lr_schedule(1.0), eps=old.defaults.get("eps", 1e-5),
                           betas=old.defaults.get("betas", (0.9, 0.999)))
            opt.load_torch_adam_state(old)
        self._hip = {"loss": loss, "opt": opt, "batch": batch, "n_mb": n_minibatches, "graph": None,
                     "n_conv": sum(p.numel() for p in self.policy.features_extractor.naive_encoder_grid.parameters())}
        self.policy.features_extractor._split_backward = self._sync is not None and self._sync.active
        if self._sync is not None and self._sync.active:
            # the rank's approx-KL rides in the slot behind the flat gradient; the flag is set from
            # the GLOBAL mean after the all-reduce (gnbv_clip_adam_step), not by the loss kernel
            loss.args.kl_out = opt.kl_slot.data_ptr()
        # one GPU: the loss launch leaves its per-sample terms behind and the optimizer's norm launch adds them up in passing (no release
        # fence + ticket per loss workgroup on the critical path); data-parallel: the KL must exist before the gradient exchange
        # (round 5: deferred there too -- gnbv_ppo_loss_finish runs on the second stream, in front of the exchange)
        loss.args.defer_stats = 1
        self.policy.features_extractor._bn_skip_flag = loss.stop_flag
        # (GENNBV_FORCE_SHARD=1: also with a one-rank communicator -- the captured reduce-scatter / all-gather code path on one GPU)
        if (self._sync is not None and self._sync.active and (self._sync.world > 1 or os.environ.get("GENNBV_FORCE_SHARD") == "1")
                and getattr(self, "shard_update", True)
                and getattr(opt, "shard", None) is None and getattr(self.policy.features_extractor, "backend", "") == "hip"):
            import torch.distributed as dist
            sl = opt.slice_of(self.policy.features_extractor.output_layer_grid[0].weight)
            if sl is not None and sl[0] == self._hip["n_conv"]:
                opt.enable_shard(sl[0], sl[1], self._sync.rank(), self._sync.world)
        if self._sync is not None and self._sync.active and self._sync.world > 1:  # (one rank: its statistics ARE the global ones)
            # global-minibatch statistics (gennbv_amd/parallel.py): advantage mean / std and BatchNorm-1's input
            # autocorrelation total come from per-train() tables (one row per minibatch, copied into these two buffers
            # before each step); BatchNorm-2 and the backward sums are summed over the ranks inside the encoder calls
            buf = self.rollout_buffer
            if buf.autocorr is None or buf.grid_i8 is None:
                raise ValueError("data-parallel training runs on the fused gfx950 path: it needs the int8 grid rows with their "
                                 "autocorrelation rows (an env with supports_grid_i8, G % 16 == 0; e.g. compact_obs=True)")
            cb, sync_buf = self._sync.encoder_sync(self.device)
            # (round 5: the two slots ARE the tail of the loss op's row buffer -- [rows | advantage statistics | autocorrelation total] --, so
            # that the rotation table of the replayed graph can deal them out with the row numbers; eager steps copy into them as before)
            self._hip["adv_cur"] = loss.adv_slot
            self._hip["ac_cur"] = loss.ac_slot
            loss.args.adv_norm = self._hip["adv_cur"].data_ptr()
            self.policy.features_extractor._dp_sync = {"world": self._sync.world, "cb": cb, "sync_buf": sync_buf,
                                                       "autocorr_global": self._hip["ac_cur"]}
        else:
            self.policy.features_extractor._dp_sync = None
        from ..ops import direct_grad
        direct_grad.enable(self.policy, self.grad_write_through)
        # with write-through every gradient slice is overwritten each step: no zero-fill needed when the
        # linears + the conv stack cover ALL trainable parameters
        covered = set()
        for m in self.policy.modules():
            if isinstance(m, torch.nn.Linear) and m.bias is not None:
                covered.update((id(m.weight), id(m.bias)))
        covered.update(id(p) for p in self.policy.features_extractor.naive_encoder_grid.parameters())
        from ..ops import encoder_ops
        from .policies import _IdentityExtractor
        enc = self.policy.features_extractor
        self._hip["fused_head"] = (getattr(enc, "backend", "") == "hip"
                                   and isinstance(self.policy.mlp_extractor, _IdentityExtractor)
                                   and encoder_ops.policy_head_supported(enc, self.policy.action_net, self.policy.value_net))
        # fc_grid's weight gradient on a second stream (only the optimizer needs it; joined in _hip_minibatch_body).  Data-parallel
        # (round 5): the exchange of the late gradients is what waits for that stream, not the conv backward.
        if getattr(enc, "backend", "") == "hip":
            enc.output_layer_grid[0]._async_wgrad = bool(self.grad_write_through) and getattr(self, "async_wgrad", True)
        self._hip["skip_zero"] = bool(self.grad_write_through) and all(
            id(p) in covered for p in self.policy.parameters() if p.requires_grad)
        # fc_grid's weight gradient (94 % of all parameters) leaves its GEMM with sum(dW^2) as fp64 partial sums: the clip's norm
        # pass skips that slice.  One GPU only: data-parallel ranks clip the all-reduced gradient, whose norm nobody has yet.
        self._hip["sq_slice"] = None
        if getattr(enc, "backend", "") == "hip":
            lin = enc.output_layer_grid[0]
            lin._dw_sq_partial, lin._dw_sq_written = None, False
            sl = opt.slice_of(lin.weight)
            if bool(self.grad_write_through) and (self._sync is None or not self._sync.active) and sl is not None and sl[0] % 4 == 0 and sl[1] % 4 == 0:
                from .. import _lib
                parts = int(_lib.load().gnbv_linear_bwd_dw_sq_parts(int(lin.weight.shape[1])))
                lin._dw_sq_partial = torch.zeros(parts, dtype=torch.float64, device=self.device)
                self._hip["sq_slice"] = (sl[0], sl[1], lin._dw_sq_partial)
        return self._hip

    def _hip_minibatch_body(self, st, phase: str = "all"):
        """gather -> forward -> fused loss + d(logits, values) -> backward -> clip + Adam; no host sync.

        Data-parallel runs split the backward in two phases so that the all-reduce of the large
        late-layer gradients (fc_grid: 55 MB of the 58 MB at G=64) overlaps the conv-stack backward:
          phase "A": everything up to the gradients of all parameters EXCEPT the conv stack, plus
                     d loss / d (conv-stack output);
          phase "B": conv-stack backward (encoder.hip kernels) from that gradient."""
        from ..ops import encoder_ops
        from ..ops.encoder_ops import RowGather
        buf, pol, loss, opt = self.rollout_buffer, self.policy, st["loss"], st["opt"]
        if phase in ("all", "A"):
            t, n = buf.buffer_size, buf.n_envs
            obs = RowGather(buf.observations[:t].view(t * n, -1), loss.rows,
                            None if buf.grid_i8 is None else buf.grid_i8[:t].view(t * n, -1), buf.compact_state_dim,
                            None if buf.autocorr is None else buf.autocorr[:t].view(t * n, -1))
            enc = pol.features_extractor
            enc._defer_pose_backward = True  # only inside this body: it calls encoder_ops.pose_branch_backward after its backward
            if st.get("fused_head"):
                fa, fg = encoder_ops.hybrid_branches(enc, obs)
                logits, values, _ = encoder_ops.policy_head(enc, pol.action_net, pol.value_net, fa, fg)
            else:
                features = pol.extract_features(obs)
                logits = pol.action_net(features)
                values = pol.value_net(features).flatten()
            enc._defer_pose_backward = False
            loss.bind(buf)  # fused gather: the loss kernel indexes the rollout arrays through loss.rows
            d_logits, d_values = loss(logits, values)
            if not st.get("skip_zero"):
                opt.zero_grad()
            lin = getattr(enc, "output_layer_grid", [None])[0]
            if lin is not None:
                lin._defer_wgrad = True  # (only around this backward: join_async_wgrads below launches what it deferred)
            if phase == "all":
                torch.autograd.backward([logits, values], [d_logits, d_values])
                if lin is not None:
                    lin._defer_wgrad = False
                encoder_ops.pose_branch_backward(enc, self.device)  # (deferred so that the conv chain is captured first: encoder_ops.hybrid_branches)
                encoder_ops.join_async_wgrads(self.device)  # fc_grid's dW / db: second stream, beside the conv backward
                if self._sync is None or not self._sync.active:
                    sq = st.get("sq_slice") if (lin is not None and getattr(lin, "_dw_sq_written", False)) else None
                    opt.step(self.max_grad_norm, loss.stop_flag, rotate=st.get("rows_rot"), sq_slice=sq,
                             loss_finish=loss.args if loss.args.defer_stats else None)
                return
            # the forward cut the graph at the conv-stack output (enc._split_backward): this backward
            # stops at that leaf and fills the gradients of every non-conv parameter
            torch.autograd.backward([logits, values], [d_logits, d_values])
            if lin is not None:
                lin._defer_wgrad = False
            # (round 5) the pose branch's backward and fc_grid's weight gradient stay on the second stream WITHOUT a join: only the exchange
            # of the late gradients needs them (_dp_step_body orders it behind that stream), phase B needs the data gradient alone
            encoder_ops.pose_branch_backward(enc, self.device, join=False)
            encoder_ops.join_async_wgrads(self.device, join=False)
        else:  # phase "B": conv-stack backward from d loss / d (conv-stack output)
            enc = pol.features_extractor
            torch.autograd.backward([enc._grid_feats_out], [enc._grid_feats_leaf.grad])

    def _hip_minibatch_tail(self, st):
        """data-parallel tail: global KL decision + clip + Adam on the summed gradient."""
        loss, opt = st["loss"], st["opt"]
        opt.step(self.max_grad_norm, loss.stop_flag, grad_scale=1.0 / self._sync.world, kl_slot_target=loss.args.target_kl,
                 rotate=st.get("rows_rot"))

    def _dp_step_body(self, st):
        """[phase A] -> exchange of the late gradients overlapped with [phase B] -> all-reduce(KL slot + conv grads) -> clip/Adam
        tail.  Capturable: RCCL collectives are recorded into the hipGraph.  Round 5: phase A leaves the pose
        branch's backward and fc_grid's weight gradient on the second stream un-joined, the loss statistics / KL and the exchange are issued
        from that stream, and phase B starts on this one as soon as fc_grid's data gradient exists.

        Sharded (default at world > 1, `opt.shard`): fc_grid's weight (13.8 M of the 14.6 M parameters at G = 64) is exchanged as a
        REDUCE-SCATTER -- every rank receives the sum of its 1 / world of that gradient --, updated by its owner only (Adam moments
        for the shard only) and ALL-GATHERED as parameters; everything else is all-reduced and updated redundantly as before.  Same
        bytes per link as the all-reduce it replaces (that IS a reduce-scatter + all-gather), but the Adam launch -- 409 MB of HBM
        traffic per step on every rank -- shrinks to 1 / world of it for 94 % of the parameters, and the gather half of the exchange
        carries parameters the next forward needs ~0.1 ms later instead of gradients the update needs at once.  The clip factor needs
        sum(g^2) of the WHOLE summed gradient: each rank adds its shard's squared sum to one fp64 that rides a 1-element all-reduce."""
        opt, sync = st["opt"], self._sync
        n_conv = st["n_conv"]
        sh = getattr(opt, "shard", None)
        self._hip_minibatch_body(st, "A")
        # the exchange of the late gradients is issued behind the SECOND stream (pose branch backward, fc_grid's weight gradient) and behind
        # what phase A left on this one (heads, fc_grid's bias): the conv backward below starts as soon as its data gradient exists
        from ..ops import encoder_ops
        assert self.device.type == "cuda"
        side = encoder_ops.second_stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        late = torch.cuda.stream(side)
        # (second stream: pose branch backward -> fc_grid's weight gradient -> statistics / KL -> the exchange is issued.  Issuing the
        # reduce-scatter right behind the weight gradient, BEFORE the pose branch's backward, was measured: 804 -> 860 ms per iteration at
        # one rank -- the collective's stream then runs beside both other streams, and a replayed graph that is three branches wide is
        # serialised by the executor, profiles/r05_notes.md sections 3 and 9)
        with late:
            # the minibatch's statistics row and this rank's approx-KL (the slot in front of the flat gradient, all-reduced with the conv
            # gradients below): one small launch beside the conv backward instead of a release fence + ticket per workgroup in
            # k_ppo_fused, on the critical path.  The main stream waits for it only AFTER phase B (`kl_ready`).
            st["loss"].finish_stats()
            stats_done = torch.cuda.Event()
            stats_done.record(side)

        def kl_ready():
            torch.cuda.current_stream(self.device).wait_event(stats_done)
        if sh is None:
            with late:
                work = sync.all_reduce(opt.grads_with_slot[opt.SLOT + n_conv:], async_op=True)
            self._hip_minibatch_body(st, "B")
            kl_ready()
            sync.all_reduce(opt.grads_with_slot[:opt.SLOT + n_conv])
            work.wait()
            self._hip_minibatch_tail(st)
            return
        lo, hi, loss = sh["lo"], sh["hi"], st["loss"]
        assert lo == n_conv, "the sharded slice is the first of the late gradients (parameter order: conv stack, fc_grid.weight, ...)"
        with late:
            w_rs = sync.reduce_scatter(sh["grad"], opt.grads[lo:hi], async_op=True)
            w_ar = sync.all_reduce(opt.grads[hi:], async_op=True)
        self._hip_minibatch_body(st, "B")
        kl_ready()
        sync.all_reduce(opt.grads_with_slot[:opt.SLOT + n_conv])
        w_rs.wait()
        # (the shard's square sum and its 2 KB all-reduce behind the reduce-scatter on the second stream, beside the conv backward, would take one
        # launch and one collective's latency off this tail: the one-rank RCCL capture of that order killed the process -- round 5, not pursued)
        opt.shard_sq()  # (one launch: 256 fp64 partial sums of the shard's squares; three torch kernels and 2 x 110 MB of fp64 temporaries before)
        sync.all_reduce(sh["sq"])
        w_ar.wait()
        opt.step(self.max_grad_norm, loss.stop_flag, grad_scale=1.0 / self._sync.world, kl_slot_target=loss.args.target_kl,
                 sq_slice=(lo, hi, sh["sq"]), skip_update=True, rotate=st.get("rows_rot"))
        opt.shard_step(loss.stop_flag)
        p_shard, _, _ = opt.shard_views()
        sync.all_gather(opt.params[lo:hi], p_shard)

    def _dp_minibatch(self, st, use_graph: bool):
        """One data-parallel optimizer step (see _dp_step_body)."""
        import torch.distributed as dist
        g = st["graph"] if use_graph else None
        if g is None:
            if self.dp_stress_spin_cycles:
                # replay-order stress (tests): the device is held back at the head of every eager step, so the host enqueues the WHOLE step --
                # both streams, every allocation and free -- before the first kernel runs, as a graph replay does.  A block handed to a second
                # stream without record_stream / an event then shows as wrong numbers here too, not only in the replayed RCCL graph.
                torch.cuda._sleep(int(self.dp_stress_spin_cycles))
            return self._dp_step_body(st)
        return g.replay()  # everything, collectives included, in one hipGraph

    def _train_hip(self) -> None:
        """train() on the gfx950 kernels (`_train_hip_once`), made safe against the operand ranges of the split-f16 arithmetic: the
        parameter pre-check moves the encoder to the fp32-MFMA kernels BEFORE anything is computed; the activation flags the kernels
        raise are only known AFTER the call, when every Adam step and BatchNorm update has been applied -- so the update state
        (flat parameters, Adam moments, step counter, module buffers, `_n_updates`: ~0.18 GB, one device copy per call) is snapshotted
        first, and a flagged call is REPEATED on the fp32-MFMA kernels from that snapshot instead of aborting learn() mid-run with
        possibly clamped results applied.  Data-parallel: the flag is the maximum over the ranks, so every rank repeats together."""
        enc = self.policy.features_extractor
        guarded = getattr(enc, "backend", "") == "hip" and hasattr(enc, "check_operand_ranges") and self.device.type == "cuda"
        snap = self._snapshot_update_state() if guarded and not getattr(enc, "force_fp32", False) else None
        self._train_hip_once()
        if not guarded:
            return
        flag = int(enc.check_operand_ranges(raise_on_flag=False)["flag"])
        if self._sync is not None and self._sync.active and self._sync.world > 1:
            import torch.distributed as dist
            t = torch.tensor([flag], dtype=torch.int32, device=self.device)
            self._sync.all_reduce_eager_(t, op=dist.ReduceOp.MAX)  # (eager, once per train(): the side group when the main one is RCCL)
            flag = int(t.item())
        self.logger.record("train/range_replays", getattr(self, "range_replays", 0))
        self.logger.record("train/encoder_fp32_kernels", int(bool(getattr(enc, "force_fp32", False))))
        if not flag:
            return
        if snap is None:  # already on the fp32 kernels: only a feature above 1000 can get here, and nothing clamps there
            return
        if (self._hip or {}).get("force_fp32"):
            # the PARAMETER pre-check inside the pass already moved the encoder to the fp32-MFMA kernels before anything was computed
            # (the snapshot was taken before that): the pass ran exact arithmetic, a second one would repeat it for nothing
            enc.check_operand_ranges(raise_on_flag=False)  # (clears the flag)
            return
        import warnings
        warnings.warn(f"[gennbv_amd] train(): an activation left the split-f16 operand range (flag {flag}); the call is repeated on the "
                      "fp32-MFMA kernels from the state it started with (exact, slower); the encoder stays on them")
        enc.force_fp32 = True
        enc.check_operand_ranges(raise_on_flag=False)  # (marks the linears `_fp32_arith`, clears the flag)
        self._restore_update_state(snap)
        self.range_replays = getattr(self, "range_replays", 0) + 1
        self._train_hip_once()
        enc.check_operand_ranges(raise_on_flag=False)

    def _snapshot_update_state(self):
        opt = self._hip["opt"] if self._hip else None
        st = {"n_updates": self._n_updates, "buffers": [b.detach().clone() for b in self.policy.buffers()]}
        if opt is not None:
            st["flat"] = [t.clone() for t in (opt.params, opt.exp_avg, opt.exp_avg_sq, opt.step_count)]
        else:  # first call: the flat optimizer does not exist yet (it is built from the torch Adam's state, which this call does not touch)
            st["params"] = [p.detach().clone() for p in self.policy.parameters()]
        return st

    def _restore_update_state(self, st) -> None:
        opt = self._hip["opt"]
        with torch.no_grad():
            for b, v in zip(self.policy.buffers(), st["buffers"]):
                b.copy_(v)
            if "flat" in st:
                for t, v in zip((opt.params, opt.exp_avg, opt.exp_avg_sq, opt.step_count), st["flat"]):
                    t.copy_(v)
            else:
                for p, v in zip(self.policy.parameters(), st["params"]):
                    p.copy_(v)  # (parameters are views of opt.params by now: this restores the flat buffer)
                opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_count.zero_()
                opt.load_torch_adam_state(self.policy.optimizer)
        self._n_updates = st["n_updates"]
        self._hip["graph"] = None  # the kernel choice is baked into the captured graph

    def _train_hip_once(self) -> None:
        """One pass of train(): same arithmetic as the reference loop (ppo_grid_obs.py:196-275), zero host synchronisation inside an
        epoch."""
        training_start = time.time()
        buf = self.rollout_buffer
        total = buf.buffer_size * buf.n_envs
        batch = int(self.batch_size)
        assert total % batch == 0, "the fused train path needs n_steps*n_envs to be a multiple of batch_size"
        n_mb = total // batch
        st = self._hip
        if st is None or st["batch"] != batch or st["n_mb"] != n_mb:
            st = self._hip_setup(batch, n_mb)
        loss, opt = st["loss"], st["opt"]
        self.policy.set_training_mode(True)
        # operand ranges of the split-f16 kernels: parameters outside them move the encoder to the fp32-MFMA kernels BEFORE anything
        # is computed (the captured graph bakes the kernel choice in: re-capture when it flips)
        fp32_now = self._check_ranges()
        if st.get("force_fp32") != fp32_now:
            st["graph"], st["force_fp32"] = None, fp32_now
        from types import SimpleNamespace
        c = SimpleNamespace(buf=buf, loss=loss, opt=opt, n_mb=n_mb, batch=batch, adv_tab=None, ac_tab=None)
        self._train_call_tables(st, c)
        self._train_call_run(st, c)
        self._train_call_log(st, c, training_start)

    def _train_call_tables(self, st, c) -> None:
        """Per-call state of the fused train(): hyper-parameters (kernel arguments: a change drops the captured graph), the row numbers of
        every minibatch, and -- for the replayed graph on one GPU -- the rotation table [rows | advantage statistics | autocorrelation
        total] the Adam launch deals out minibatch by minibatch."""
        c.lr = self.lr_schedule(self._current_progress_remaining)
        self.logger.record("train/learning_rate", c.lr)
        c.opt.lr = c.lr
        c.clip_range = self.clip_range(self._current_progress_remaining)
        c.clip_range_vf = None if self.clip_range_vf is None else self.clip_range_vf(self._current_progress_remaining)
        c.loss.args.clip_range = float(c.clip_range)
        c.loss.args.clip_range_vf = float(c.clip_range_vf) if c.clip_range_vf is not None else -1.0
        c.loss.stats_row.zero_()
        c.loss.stop_flag.zero_()
        idx = torch.from_numpy(np.asarray(c.buf.indices, dtype=np.int64)).to(self.device)
        c.rows_all = c.buf.rows_of(idx)  # the reference's flattened index -> row of the [T, N] layout
        c.use_graph = self.use_graph and self.device.type == "cuda" and not st.get("graph_refused")
        hyper = (float(c.lr), float(c.clip_range), None if c.clip_range_vf is None else float(c.clip_range_vf))
        if st.get("hyper") != hyper:
            st["graph"], st["hyper"] = None, hyper  # kernel arguments are baked into the graph: re-capture
        c.dp = self._sync is not None and self._sync.active
        c.dp_stats = c.dp and self._sync.world > 1
        if c.dp_stats:
            # the advantages and the permutation are fixed for the whole train() call: the global minibatches' advantage
            # statistics and input-autocorrelation totals are computed once (three small all-reduces), not per step
            t_, n_ = c.buf.buffer_size, c.buf.n_envs
            c.adv_tab = self._sync.global_adv_norm(c.buf.advantages.view(t_ * n_)[c.rows_all].view(c.n_mb, c.batch))
            c.ac_tab = self._sync.global_autocorr(c.buf.autocorr[:t_].view(t_ * n_, -1)[c.rows_all].view(c.n_mb, c.batch, -1))
            st["adv_cur"].copy_(c.adv_tab[0])
            st["ac_cur"].copy_(c.ac_tab[0])
        # Replayed graph on one GPU: the row numbers of ALL minibatches of this call go to a table once, and the Adam launch that ends
        # a minibatch leaves the next one's in `loss.rows` (gnbv_clip_adam_step_rotate) -- no copy and no host work between two
        # replays.  (The table and the counter are baked into the graph: persistent buffers.)
        # Data-parallel (round 5): the same table when the step is one hipGraph (RCCL) -- its statistics columns then hold the GLOBAL
        # minibatches' figures computed above; with eager collectives (gloo) the three slots are copied between two steps as before.
        c.rotating = (c.use_graph and c.n_mb > 0 and self.rotate_rows
                      and (not c.dp or self._collectives_capturable()))
        c.rot = st.get("rows_rot")
        if c.rotating and (c.rot is None or tuple(c.rot[0].shape) != (c.n_mb, c.batch + 1 + 384)):
            # a table row = [the minibatch's row numbers | (mean, 1 / (std + 1e-8)) of its advantages | the sum of its input-autocorrelation
            # rows], rotated into loss.rows_ext
            c.rot = (torch.zeros(c.n_mb, c.batch + 1 + 384, dtype=torch.int64, device=self.device), c.loss.rows_ext, torch.zeros(1, dtype=torch.int32, device=self.device))
            st["rows_rot"], st["graph"] = c.rot, None
        elif not c.rotating and c.rot is not None:
            c.rot = st["rows_rot"] = None
            st["graph"] = None
        if not c.dp_stats:  # (several ranks: always the slot, filled from the global table -- _hip_setup)
            c.loss.args.adv_norm = c.loss.adv_slot.data_ptr() if (c.rotating and self.normalize_advantage) else None
        if not c.rotating:
            self.policy.features_extractor._autocorr_total = None
            if st.get("ac_total_on"):
                st["graph"], st["ac_total_on"] = None, False
        if c.rotating:
            c.rot[0][:, :c.batch].copy_(c.rows_all[:c.n_mb * c.batch].view(c.n_mb, c.batch))
            if c.dp_stats:
                c.rot[0][:, c.batch:c.batch + 1].view(torch.float32).copy_(c.adv_tab)
                c.rot[0][:, c.batch + 1:].view(torch.int32).copy_(c.ac_tab)
            elif self.normalize_advantage:
                # The advantages and the permutation are fixed for the whole train() call: every minibatch's statistics
                # (ppo_grid_obs.py:214-216: mean, unbiased std) once, instead of two dependent gather passes in every wave of every
                # loss launch
                adv = c.buf.advantages.view(-1)[c.rows_all[:c.n_mb * c.batch]].view(c.n_mb, c.batch)
                stats = torch.stack((adv.mean(1), 1.0 / (adv.std(1) + 1e-8)), 1).contiguous()
                c.rot[0][:, c.batch:c.batch + 1].view(torch.float32).copy_(stats)
            # BatchNorm-1's batch statistics come from the SUM of the minibatch's autocorrelation rows: one table per train() call
            # instead of a gather of 128 scattered rows in front of every forward (k_bn1_analytic)
            enc_ = self.policy.features_extractor
            use_tot = c.buf.autocorr is not None and not c.dp_stats  # (several ranks: GnbvEncoderParams.autocorr_global = the same slot)
            if use_tot:
                ac_rows = c.buf.autocorr[:c.buf.buffer_size].view(c.buf.buffer_size * c.buf.n_envs, -1)
                tot = ac_rows[c.rows_all[:c.n_mb * c.batch]].view(c.n_mb, c.batch, -1).sum(1, dtype=torch.int64)
                c.rot[0][:, c.batch + 1:].view(torch.int32).copy_(tot.to(torch.int32))
            if st.get("ac_total_on") != use_tot:
                st["graph"], st["ac_total_on"] = None, use_tot  # (the pointer is a kernel argument baked into the graph)
            enc_._autocorr_total = c.loss.ac_slot if use_tot else None  # (only for the duration of this call: cleared below)
            c.rot[2].zero_()

    def _train_call_run(self, st, c) -> None:
        """Capture (when the graph was dropped) and the epochs x minibatches loop: no host synchronisation inside an epoch."""
        st["replays_per_call"] = c.n_mb * self.n_epochs
        st["calls_since_capture"] = st.get("calls_since_capture", 0) + 1
        if c.use_graph and st["graph"] is None:
            if c.rotating:
                c.loss.rows_ext.copy_(c.rot[0][0])
            else:
                c.loss.rows.copy_(c.rows_all[:c.batch])
            st["graph"] = self._capture_minibatch_graph(st)
            if st["graph"] is None:  # (data-parallel only: the collectives could not be captured -> eager steps from here on)
                st["graph_refused"], c.use_graph = True, False
            c.loss.stats_row.zero_()
            c.loss.stop_flag.zero_()
            if c.rotating:
                c.rot[2].zero_()
        if c.rotating:
            c.loss.rows_ext.copy_(c.rot[0][0])
        try:
            for epoch in range(self.n_epochs):
                for k in range(c.n_mb):
                    if not c.rotating:
                        c.loss.rows.copy_(c.rows_all[k * c.batch:(k + 1) * c.batch])
                    if c.dp_stats and not c.rotating:
                        st["adv_cur"].copy_(c.adv_tab[k])
                        st["ac_cur"].copy_(c.ac_tab[k])
                    if c.dp:
                        self._dp_minibatch(st, c.use_graph)
                    elif c.use_graph:
                        st["graph"].replay()
                    else:
                        self._hip_minibatch_body(st)
                # the ONLY read-back inside train(): early-stop flag, once per epoch (the reference
                # reads approx_kl on the host after every minibatch, :261-268)
                if self.target_kl is not None and int(c.loss.stop_flag.item()) != 0:
                    if self.verbose >= 1:
                        print(f"Early stopping at step {epoch} due to reaching max kl")
                    break
        finally:
            # (also when the loop raises: the slot holds the LAST minibatch's total -- never for another caller's training-mode forward)
            self.policy.features_extractor._autocorr_total = None

    def _train_call_log(self, st, c, training_start) -> None:
        """The call's only large read-back: the statistics table -> the reference's logger records (ppo_grid_obs.py:277-292)."""
        self._n_updates += self.n_epochs
        if c.dp and getattr(c.opt, "shard", None) is not None and self._sync.world > 1:
            # sharded fc_grid update: the owners' Adam moments into every rank's flat buffers HERE, at a point every rank passes together
            # (two all-gathers of 55 MB per train() call), so that get_parameters() / save() never need a collective
            self._gather_shard_state(st, c.opt)
        rows_done = int(c.loss.stats_row.item())
        s = c.loss.stats[:rows_done].double().cpu().numpy()
        s = s[s[:, 6] > 0.5]  # minibatches the reference would have executed
        self.last_train_stats = s
        last_epoch = (len(s) - 1) // c.n_mb
        v_flat, r_flat = c.buf.flat_values_returns()
        var_y = torch.var(r_flat, unbiased=False)
        explained_var = float("nan") if float(var_y) == 0 else float(1 - torch.var(r_flat - v_flat, unbiased=False) / var_y)
        self.logger.record("train/entropy_loss", float(np.mean(s[:, 2])))
        self.logger.record("train/policy_gradient_loss", float(np.mean(s[:, 0])))
        self.logger.record("train/value_loss", float(np.mean(s[:, 1])))
        self.logger.record("train/approx_kl", float(np.mean(s[last_epoch * c.n_mb:, 3])))
        self.logger.record("train/clip_fraction", float(np.mean(s[:, 4])))
        self.logger.record("train/loss", float(s[-1, 5]))
        self.logger.record("train/explained_variance", explained_var)
        self.logger.record("train/n_updates", self._n_updates)
        self.logger.record("train/clip_range", c.clip_range)
        if c.clip_range_vf is not None:
            self.logger.record("train/clip_range_vf", c.clip_range_vf)
        self.logger.record("time/training", time.time() - training_start)

    def _gather_shard_state(self, st, opt) -> None:
        """The owners' Adam moments into every rank's flat buffers at the end of train().  With RCCL the two all-gathers are a captured
        hipGraph of their own (captured once per optimizer state, replayed per call): the communicator then carries captured work only
        (VERDICT r5 item 5a); backends whose collectives cannot be captured run them eagerly."""
        if not (self.use_graph and self.device.type == "cuda" and not st.get("graph_refused") and self._collectives_capturable()):
            return opt.gather_shard_state(self._sync.group)
        g = st.get("gather_graph")
        if g is None:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                opt.gather_shard_state(self._sync.group)
            st["gather_graph"] = g
        g.replay()

    def _check_ranges(self) -> bool:
        """Hybrid_Encoder.check_operand_ranges (split-f16 kernels' limits made loud) -> whether the encoder is on the fp32 kernels."""
        enc = self.policy.features_extractor
        if getattr(enc, "backend", "") != "hip" or not hasattr(enc, "check_operand_ranges"):
            return False
        return bool(enc.check_operand_ranges()["force_fp32"])

    def _capture_minibatch_graph(self, st):
        """Capture gather+forward+loss+backward+Adam of one minibatch as ONE hipGraph (data-parallel with RCCL: the collectives are
        recorded into it; a backend whose collectives cannot be captured gets None = eager launches of the same step).  Warm-up runs
        happen on a side stream with the update masked (stop_flag = 1), so parameters, Adam state and BatchNorm running statistics
        are untouched."""
        loss = st["loss"]
        dp = self._sync is not None and self._sync.active
        # Data-parallel: the eager warm-up (its collectives are the only eager work the RCCL communicator ever sees besides the rendezvous)
        # runs in front of the FIRST capture of an optimizer state only; re-captures (a learning-rate / clip-range schedule re-captures in
        # every train() call) find kernels, workspaces and the communicator warm.
        if not (dp and getattr(self, "_dp_warm_for", None) is st["opt"]):
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    loss.stop_flag.fill_(1)
                    if dp:
                        self._dp_step_body(st)
                    else:
                        self._hip_minibatch_body(st)
            torch.cuda.current_stream(self.device).wait_stream(side)
            if dp:
                self._dp_warm_for = st["opt"]
                self._dp_eager_since_sync = True
        loss.stop_flag.fill_(1)
        # thread_local: the RCCL watchdog thread may touch the HIP runtime while we capture
        ga = torch.cuda.CUDAGraph()
        if not dp:
            # The memory pool of the capture that was kept is PINNED (st["graph_pool"]): a later re-capture (every invalidation above drops
            # the old graph first, so the pool's blocks are free) allocates the same intermediates at the same addresses.  Why: what makes
            # one capture replay 5-10 us slower than another for its whole life is WHERE its private pool put the minibatch's
            # intermediates (y1 2 x 242 MB, four 28 MB tensors) -- captures alive together, at different addresses, keep their rank over
            # rounds of alternating replays, sequential re-captures into the same blocks agree within 2-5 us, and graphs WITHOUT the
            # second stream show the same spread (tools/capture_states.py, profiles/r05_capture_states*.txt): not the executor's queue
            # placement, as round 4 assumed.
            # (a torch.cuda.MemPool object keeps its pool alive while no graph uses it: a bare pool id dies with its last graph)
            pinned = st.get("graph_pool")
            mp = pinned if pinned is not None else torch.cuda.MemPool()
            with torch.cuda.graph(ga, pool=mp.id, capture_error_mode="thread_local"):
                self._hip_minibatch_body(st)
            keep, st["graph_pool"] = self._best_of_captures(st, ga, mp, fixed_placement=pinned is not None)
            return keep
        if not self._collectives_capturable():
            # Fall back to the EAGER data-parallel step (same `_dp_step_body`, same sharded update, launch by launch): the compute cannot be
            # captured by itself either -- BatchNorm's batch sums are exchanged inside the encoder calls (GnbvEncoderParams.sync_sum), so
            # every piece of the step contains a collective.  (Rounds 2-3 fell back to "two compute graphs + eager collectives"; the
            # first 2-rank test of that path, round 4, showed it cannot work: the second capture dies on the same collective.)
            loss.stop_flag.zero_()
            return None
        # the eager collectives above (warm-up steps; attach()'s broadcasts before them) are finished AND retired by the process group's
        # watchdog thread (it polls every 100 ms) before the first captured collective records an event: see parallel.capture_safe_env
        torch.cuda.synchronize(self.device)
        if getattr(self, "_dp_eager_since_sync", False) and self.dp_capture_settle_s > 0:
            time.sleep(self.dp_capture_settle_s)  # (only behind eager collectives, i.e. in front of the first capture)
        self._dp_eager_since_sync = False
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            self._dp_step_body(st)
        self.dp_graph_mode = "one hipGraph incl. RCCL collectives"
        return ga

    def _best_of_captures(self, st, first, first_pool=None, fixed_placement=False):
        """A captured minibatch lands in one of several states PER CAPTURE -- the same kernels replay at 507-515 or at 521-532 us,
        stable for the life of the graph object (round 5: the state is the PLACEMENT of the capture's private memory pool, see
        _capture_minibatch_graph; the candidates below are candidates for a placement, and the winner's pool is pinned for every
        later re-capture).  A train() call of BASELINE configs[1] replays the graph 1280 times, so when
        the call is long enough to pay for it, the step is captured `graph_candidates` times and the fastest capture kept: each candidate is
        replayed with the update masked (stop_flag = 1: parameters, Adam state and BatchNorm statistics untouched, as in the warm-up runs),
        timed with events; the others are dropped with their memory pools."""
        k = self.graph_candidates
        if fixed_placement and k is None:
            k = 1  # (a re-capture into the pinned pool: the placement was chosen when the pool was)
        if k is None:
            # (a learning-rate / clip-range schedule re-captures the graph in every train() call -- the hyper-parameters are kernel
            # arguments --: candidates only for the first capture and for one that replaces a graph that lived >= 4 calls)
            stable = st.get("captures", 0) == 0 or st.get("calls_since_capture", 0) >= 4
            k = 3 if (st.get("replays_per_call", 0) >= 256 and stable) else 1
        st["captures"], st["calls_since_capture"] = st.get("captures", 0) + 1, 0
        if k <= 1:
            return first, first_pool
        loss = st["loss"]
        cands, pools = [first], [first_pool]
        for _ in range(k - 1):
            loss.stop_flag.fill_(1)
            g, mp = torch.cuda.CUDAGraph(), torch.cuda.MemPool()
            with torch.cuda.graph(g, pool=mp.id, capture_error_mode="thread_local"):
                self._hip_minibatch_body(st)
            cands.append(g)
            pools.append(mp)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = [[] for _ in cands]
        steps_before = int(st["opt"].step_count.item())  # (the masked replays below must not count as optimizer steps: checked after the loop)
        for rnd in range(3):  # alternate the candidates: clock / thermal drift is common to them
            for c, g in enumerate(cands):
                loss.stop_flag.fill_(1)
                for j in range(13):
                    if j == 3:
                        ev0.record()
                    loss.stats_row.zero_()  # (every replay appends a statistics row: the table only holds one train() call)
                    g.replay()
                ev1.record()
                ev1.synchronize()
                times[c].append(ev0.elapsed_time(ev1) / 10.0)
        if int(st["opt"].step_count.item()) != steps_before:
            raise RuntimeError("a masked replay (stop_flag = 1) advanced the optimizer: a kernel of the captured minibatch ignores the stop flag")
        med = [sorted(t)[1] for t in times]
        best = min(range(len(cands)), key=lambda c: med[c])
        self.graph_capture_ms = [round(m, 4) for m in med]  # (bench.py reports it)
        keep, keep_pool = cands[best], pools[best]
        del cands, pools
        return keep, keep_pool

    def _collectives_capturable(self) -> bool:
        """Can this process group's collectives be recorded into a hipGraph?  RCCL (backend "nccl"): yes -- the step, collectives
        included, is one graph.  Anything else (gloo on device tensors: the multi-rank tests on one GPU) synchronises the stream inside
        the collective, which a capture refuses -- and a capture refused half-way cannot be cleaned up from Python (torch's
        `capture_end` raises before it restores the current stream, the stream stays `invalidated`, later collectives fail from the
        autograd thread: tried in round 4), so the question is answered from the backend's name BEFORE anything is captured."""
        ok, backend = self._sync.capturable()
        if not ok:
            self.dp_graph_mode = f"eager launches (the {backend} backend's collectives cannot be captured)"
        return ok

    # ------------------------------------------------------------------------------
    def _env_step(self, actions, obs_out, defer_autocorr: bool = False):
        g8 = self.rollout_buffer.next_grid_i8_row()
        if g8 is not None:
            out = self.env.step(actions, obs_out=obs_out, grid_i8_out=g8)
            if not defer_autocorr:  # (deferred: the caller issues it behind the policy evaluation's second-stream work -- collect_rollouts)
                self.rollout_buffer.update_autocorr(self.rollout_buffer.step + 1)
            return out
        try:
            return self.env.step(actions, obs_out=obs_out)
        except TypeError:
            return self.env.step(actions)

    def _with_grid_i8(self, obs, row: int):
        """The rollout forward reads the compact grid copy of buffer row `row` when there is one (fused path only)."""
        buf = self.rollout_buffer
        from ..ops.encoder_ops import DenseObs
        if buf.compact_state_dim is not None:  # compact rows: the grid exists only in the int8 rows
            assert obs.data_ptr() == buf.observations[row].data_ptr(), "compact observations live in the rollout buffer"
            return DenseObs(obs, buf.grid_i8[row], buf.compact_state_dim)
        if buf.grid_i8 is None or not getattr(self.policy, "_fused_rollout", False) or obs.data_ptr() != buf.observations[row].data_ptr():
            return obs
        return DenseObs(obs, buf.grid_i8[row])

    def _maybe_enable_grid_i8(self) -> None:
        """Compact int8 copy of the tri-class grid rows (written by the env's coded state-encoding kernel, read by the
        conv1 kernels of the update): on when the env offers it, the encoder runs on the gfx950 kernels and G % 16 == 0."""
        enc = self.policy.features_extractor
        g = getattr(enc, "grid_size", 0)
        if (self.grid_i8_rows and getattr(self.env, "supports_grid_i8", False)
                and getattr(enc, "backend", "") == "hip" and g % 16 == 0 and self.rollout_buffer.grid_i8 is None):
            self.rollout_buffer.enable_grid_i8(g ** 3)

    def _refresh_grid_i8_row0(self) -> None:
        """row 0 of the int8 copy from the fp32 observation row 0 (only when that row was not written by the env)."""
        buf, enc = self.rollout_buffer, self.policy.features_extractor
        if buf.grid_i8 is not None and buf.compact_state_dim is None:
            s0 = enc.state_input_shape[0]
            buf.grid_i8[0].copy_(buf.observations[0][:, s0:s0 + enc.grid_size ** 3].to(torch.int8))
            buf.update_autocorr(0)

    def collect_rollouts(self, env, callback, rollout_buffer, n_rollout_steps: int) -> bool:
        """on_policy_algorithm_grid_obs.py:128-221 (tensor-env branch).

        Inside the rollout the env hands out `dones` / `infos["time_outs"]` as views of its kernels' bytes (`flag_views`: no `.bool()`
        launches); both are consumed before the env overwrites them.  The switch is put back however the rollout ends (early return of a
        callback, exception), and what outlives the call -- `_last_episode_starts` -- is a copy, so env steps taken between two rollouts
        (an evaluation on the same env, user code after learn()) see the reference's fresh-tensor behaviour and cannot reach into the
        next rollout's row 0.  Callbacks that KEEP `self.locals["dones"]` beyond their `on_step()` must clone it.
        """
        prev = getattr(env, "flag_views", None)
        if prev is not None:
            env.flag_views = True
        try:
            return self._collect_rollouts(env, callback, rollout_buffer, n_rollout_steps)
        finally:
            if prev is not None:
                env.flag_views = prev
                if not prev and self._last_episode_starts is not None:
                    self._last_episode_starts = self._last_episode_starts.clone()

    def _collect_rollouts(self, env, callback, rollout_buffer, n_rollout_steps: int) -> bool:
        assert self._last_obs is not None, "No previous observation was provided"
        self.policy.set_training_mode(False)
        if not hasattr(self.policy, "_fused_rollout"):
            from ..ops import encoder_ops
            from .policies import _IdentityExtractor
            enc = self.policy.features_extractor
            self.policy._fused_rollout = (
                self.device.type == "cuda"
                and getattr(enc, "backend", "") == "hip" and isinstance(self.policy.mlp_extractor, _IdentityExtractor)
                and hasattr(self.policy.action_dist, "sample_and_log_prob")
                and encoder_ops.policy_head_supported(enc, self.policy.action_net, self.policy.value_net))
        fused_add = (self.device.type == "cuda" and getattr(self.policy, "_fused_rollout", False)
                     and self.fused_add)
        n_steps = 0
        self._check_ranges()
        # the policy evaluation of this rollout's steps with its step-invariant work hoisted out (ops/rollout_plan.py); None: general path
        plan = self._rollout_forward(env.num_envs) if fused_add or getattr(self.policy, "_fused_rollout", False) else None

        # The new row's input autocorrelation (BatchNorm-1's analytic statistics in train(): nothing in the rollout reads it) goes to the
        # second stream BEHIND the pose branch of the policy evaluation, where it runs beside the conv kernel instead of in front of it.
        defer_ac = plan is not None and rollout_buffer.autocorr is not None

        def evaluate(x, values_only=False, autocorr_row=None):
            if plan is not None and plan.applies_to(x):
                tail = None if autocorr_row is None else (lambda st, r=autocorr_row: rollout_buffer.update_autocorr(r, stream=st))
                logits, v = plan(x, tail)
                if values_only:
                    return v.unsqueeze(1)
                a, lp = self.policy.action_dist.sample_and_log_prob(logits, False)
                return a, v.unsqueeze(1), lp
            if autocorr_row is not None:
                rollout_buffer.update_autocorr(autocorr_row)
            return self.policy.predict_values(x) if values_only else self.policy(x)
        rollout_buffer.reset()
        first = rollout_buffer.first_obs_row()
        if self._last_obs.data_ptr() != first.data_ptr():
            first.copy_(self._last_obs)
            self._last_obs = first
            self._refresh_grid_i8_row0()
        if callback is not None:
            callback.on_rollout_start()
        dones = None
        new_obs = None
        while n_steps < n_rollout_steps:
            with torch.no_grad():
                if self._pending is None:
                    actions, values, log_probs = evaluate(self._with_grid_i8(self._last_obs, rollout_buffer.step))
                else:
                    actions, values, log_probs = self._pending
            new_obs, rewards, dones, infos = self._env_step(actions, rollout_buffer.next_obs_row(), defer_autocorr=defer_ac)
            self.num_timesteps += env.num_envs
            if callback is not None:
                callback.update_locals(locals())
                if callback.on_step() is False:
                    if defer_ac:  # (leave the buffers as the general path would: this step's row and the second stream joined)
                        rollout_buffer.update_autocorr(rollout_buffer.step + 1)
                        torch.cuda.current_stream(self.device).wait_stream(plan.side)
                    return False
            self._update_info_buffer(infos)
            n_steps += 1
            with torch.no_grad():
                # ONE policy evaluation of new_obs: its value is the time-out bootstrap of this
                # step (:205-208) and its action / value / log-prob are next step's (:168).
                # The last step only needs the value (:213-215) and must not draw from the RNG.
                # (a replayed hipGraph of this evaluation -- two alternating graphs over RowGather(all rows, device-side rows) -- was
                # measured in round 4: 535 against 509 us per env step; the step is not host-bound enough to pay for the graph's
                # cross-queue hand-overs.  profiles/r04_notes.md)
                new_in = self._with_grid_i8(new_obs, rollout_buffer.step + 1)  # (the buffer's step counter advances in add())
                ac_row = rollout_buffer.step + 1 if defer_ac else None
                if n_steps < n_rollout_steps:
                    nxt = evaluate(new_in, autocorr_row=ac_row)
                    terminal_value = nxt[1]
                else:
                    nxt = None
                    terminal_value = evaluate(new_in, values_only=True, autocorr_row=ac_row)
            assert self.timeout_bootstrap in ("reference", "per_env")
            first = self.timeout_bootstrap == "reference"
            if fused_add and self._last_obs.data_ptr() == rollout_buffer.observations[rollout_buffer.step].data_ptr():
                # time-out bootstrap + the five buffer copies as one launch (instead of ~9)
                rollout_buffer.add_bootstrapped(self._last_obs, actions, rewards, infos["time_outs"], terminal_value, self.gamma,
                                                self._last_episode_starts, values, log_probs, broadcast_first=first)
            else:
                tv = terminal_value[0] if first else terminal_value  # (:206: `predict_values(new_obs)[0]`)
                rewards = rewards + self.gamma * torch.squeeze(tv * infos["time_outs"].unsqueeze(1).to(self.device), 1)
                rollout_buffer.add(self._last_obs, actions, rewards, self._last_episode_starts, values, log_probs)
            self._last_obs = new_obs
            self._last_episode_starts = dones
            self._pending = nxt
        last_values = terminal_value  # V(new_obs) of the last step (:213-215)
        if plan is not None:
            torch.cuda.current_stream(self.device).wait_stream(plan.side)  # (the deferred autocorrelation rows: train() reads them)
        self._check_ranges()
        rollout_buffer.compute_returns_and_advantage(last_values=last_values, dones=dones)
        if callback is not None:
            callback.on_rollout_end()
        return True

    def _rollout_forward(self, n: int):
        """The prepared policy evaluation of this rollout (ops/rollout_plan.RolloutForward), or None (attribute `rollout_plan = False`,
        another device, a policy whose inference forward is not the kernel sequence the plan issues)."""
        if not self.rollout_plan or self.device.type != "cuda" or "forward" in vars(self.policy) or "predict_values" in vars(self.policy):
            return None  # (an instance-level override of the policy's evaluation -- tests force actions that way -- keeps the general path)
        from .policies import ActorCriticPolicy_Train_Eval as _P
        enc = self.policy.features_extractor
        if any(getattr(type(self.policy), m, None) is not getattr(_P, m) for m in ("forward", "predict_values", "_fused_head", "extract_features")):
            return None  # (a SUBCLASS that overrides the evaluation: the plan would silently bypass it)
        if any(getattr(m, h, None) for m in (self.policy, enc) for h in ("_forward_hooks", "_forward_pre_hooks")):
            return None  # (nn.Module hooks on the policy / the encoder only fire on the general path)
        from ..ops.rollout_plan import RolloutForward
        plan = getattr(self, "_rollout_plan_obj", None)
        if plan is None or plan.n != n or plan.policy is not self.policy:
            plan = self._rollout_plan_obj = RolloutForward.build(self.policy, n)
        with torch.no_grad():
            return plan if (plan is not None and plan.prepare()) else None

    def _update_info_buffer(self, infos) -> None:
        if self.ep_info_buffer is not None:
            self.ep_info_buffer.append(infos.get("episode"))

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
