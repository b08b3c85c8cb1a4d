"""collect_rollouts of PPO_Grid_Obs (stable_baselines3/common/on_policy_algorithm_grid_obs.py:128-221, tensor-env branch): one policy
evaluation per env step, observations written in place into the rollout buffer, one-launch bootstrap + add.  (Split out of
ppo_grid_obs.py in round 6.)"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F


class RolloutMixin:
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
        if (not self.rollout_plan or os.environ.get("GENNBV_ROLLOUT_PLAN") == "0" or self.device.type != "cuda" or "forward" in vars(self.policy)
                or "predict_values" in vars(self.policy)):
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
