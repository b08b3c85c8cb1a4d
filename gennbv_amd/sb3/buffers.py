"""TensorRolloutBuffer_Grid_Obs (stable_baselines3/common/buffers.py:628-762).

Same public surface (`reset`, `add`, `compute_returns_and_advantage`, `get`,
`.observations/.actions/.rewards/.values/.log_probs/.returns/.advantages/
.episode_starts`, `.indices`), redesigned for 288 GB of HBM:

  * allocated ONCE ([T+1, N, D_obs] fp32 observations; the reference re-allocates and
    zero-fills the whole buffer every rollout, :655-674); row t+1 can be handed to the
    env as the destination of its next observation (`next_obs_row`), so `add()` does
    not copy observations at all;
  * the GAE scan is one gfx950 kernel (gennbv_amd/csrc/gae.hip) instead of T x 8 launches;
  * `get()` does not materialise the swapped [N*T, D_obs] copy (:56-69,:739-740): a
    minibatch index i maps to (env n = i // T, step t = i % T) -- the same row the
    reference's swap_and_flatten puts at position i -- and rows are gathered from the
    [T, N] layout directly;
  * `indices = np.random.permutation(T*N)` is drawn in `reset()` from numpy's global RNG
    and reused by every epoch, exactly like the reference (:673, :749-751).
"""
from __future__ import annotations

from typing import Generator, NamedTuple, Optional

import numpy as np
import torch

from .. import gae as gae_ops


class RolloutBufferSamples(NamedTuple):
    observations: torch.Tensor
    actions: torch.Tensor
    old_values: torch.Tensor
    old_log_prob: torch.Tensor
    advantages: torch.Tensor
    returns: torch.Tensor


class TensorRolloutBuffer_Grid_Obs:
    def __init__(self, buffer_size: int, observation_space, action_space, device="cpu", gae_lambda: float = 1,
                 gamma: float = 0.99, n_envs: int = 1, compact=None):
        """compact = (state_dim, grid_elems): COMPACT observation storage -- `observations` rows hold only
        [state | state_rgb] in fp32 and the tri-class grid (values -1/0/1) lives solely in `grid_i8`
        ([T+1, N, G^3] int8): D_obs*4 -> (D_obs - G^3)*4 + G^3 bytes per row (1.08 MB -> 0.30 MB at G = 64)."""
        self.buffer_size, self.n_envs = int(buffer_size), int(n_envs)
        self.num_transitions_per_env, self.num_envs = self.buffer_size, self.n_envs
        self.observation_space, self.action_space = observation_space, action_space
        self.obs_shape = tuple(observation_space.shape)
        self.actions_shape = action_space.shape[0]
        self.device = torch.device(device)
        self.gae_lambda, self.gamma = gae_lambda, gamma
        t, n, dev = self.buffer_size, self.n_envs, self.device
        self.compact_state_dim = None
        if compact is not None:
            assert len(self.obs_shape) == 1
            self.compact_state_dim, self.grid_elems = int(compact[0]), int(compact[1])
            self.observations = torch.zeros(t + 1, n, self.obs_shape[0] - self.grid_elems, device=dev)
        else:
            self.observations = torch.zeros(t + 1, n, *self.obs_shape, device=dev)
        self.rewards = torch.zeros(t, n, 1, device=dev)
        self.actions = torch.zeros(t, n, self.actions_shape, device=dev)
        self.episode_starts = torch.zeros(t, n, 1, device=dev, dtype=torch.uint8)
        self.log_probs = torch.zeros(t, n, 1, device=dev)
        self.values = torch.zeros(t, n, 1, device=dev)
        self.returns = torch.zeros(t, n, 1, device=dev)
        self.advantages = torch.zeros(t, n, 1, device=dev)
        self.privileged_observations = None
        self.lazy_obs = False  # True: minibatch observations are RowGather views (gather fused into conv1)
        self.grid_i8 = None    # optional [T+1, N, G^3] int8 copy of the grid slices (enable_grid_i8)
        self.autocorr = None   # optional [T+1, N, 768] int32 (enable_grid_i8, G % 16 == 0)
        if compact is not None:
            self.enable_grid_i8(self.grid_elems)
        self.reset()

    def enable_grid_i8(self, grid_elems: int) -> None:
        """Allocate the compact copy of the tri-class grid slices: the env's state-encoding kernel fills its rows
        next to the fp32 observation rows, the conv1 kernels of the PPO update read it (a quarter of the bytes)."""
        if self.grid_i8 is None:
            self.grid_i8 = torch.zeros(self.buffer_size + 1, self.n_envs, int(grid_elems), dtype=torch.int8, device=self.device)
            # per-row autocorrelation of the conv1 input patches (3 KiB per row): a property of the stored grid, computed
            # once per env step (update_autocorr), read by the fused backward kernel of every epoch
            from ..ops import encoder_ops
            g = round(int(grid_elems) ** (1.0 / 3.0))
            if g ** 3 == int(grid_elems) and encoder_ops.autocorr_supported(g) and self.device.type == "cuda":
                self._autocorr_grid = g
                self.autocorr = torch.zeros(self.buffer_size + 1, self.n_envs, 768, dtype=torch.int32, device=self.device)

    def update_autocorr(self, row: int, stream=None) -> None:
        """(Re)compute the autocorrelation rows of observation row `row` from its int8 grid rows (`stream`: raw HIP stream, default current)."""
        if self.autocorr is not None:
            from ..ops import encoder_ops
            encoder_ops.input_autocorr(self.grid_i8[row], self._autocorr_grid, out=self.autocorr[row], stream=stream)

    def next_grid_i8_row(self):
        return None if self.grid_i8 is None else self.grid_i8[self.step + 1]

    def reset(self) -> None:
        if getattr(self, "step", 0) == self.buffer_size:
            # the observation that followed the last transition opens the next rollout
            self.observations[0].copy_(self.observations[self.buffer_size])
            if self.grid_i8 is not None:
                self.grid_i8[0].copy_(self.grid_i8[self.buffer_size])
            if self.autocorr is not None:
                self.autocorr[0].copy_(self.autocorr[self.buffer_size])
        self.step = 0
        self.pos = 0
        self.full = False
        self.generator_ready = False
        self.indices = np.random.permutation(self.buffer_size * self.n_envs)
        self._indices_dev = None

    def next_obs_row(self) -> torch.Tensor:
        """Storage of the observation that will be `add()`ed at the NEXT step."""
        return self.observations[self.step + 1]

    def first_obs_row(self) -> torch.Tensor:
        return self.observations[0]

    def add(self, obs, action, reward, episode_start, value, log_prob) -> None:
        if isinstance(episode_start, np.ndarray):
            episode_start = torch.from_numpy(episode_start).to(self.device)
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        if type(obs) is tuple:
            obs = obs[0]
        row = self.observations[self.step]
        if obs.data_ptr() != row.data_ptr():
            if self.compact_state_dim is not None and obs.shape[-1] == self.obs_shape[0]:  # a flat fp32 row: split it
                s0, ge = self.compact_state_dim, self.grid_elems
                row[:, :s0].copy_(obs[:, :s0])
                row[:, s0:].copy_(obs[:, s0 + ge:])
                self.grid_i8[self.step].copy_(obs[:, s0:s0 + ge].to(torch.int8))
                self.update_autocorr(self.step)
            else:
                row.copy_(obs)
        self.actions[self.step].copy_(action)
        self.rewards[self.step].copy_(reward.view(-1, 1))
        self.episode_starts[self.step].copy_(episode_start.view(-1, 1))
        self.values[self.step].copy_(value)
        self.log_probs[self.step].copy_(log_prob.view(-1, 1))
        self.step += 1
        self.pos += 1
        if self.pos == self.buffer_size:
            self.full = True

    def add_bootstrapped(self, obs, action, reward, time_outs, terminal_value, gamma: float, episode_start, value, log_prob,
                         broadcast_first: bool = False) -> None:
        """`rewards += gamma * squeeze(terminal_value * time_outs)` (on_policy_algorithm_grid_obs.py:205-208) followed by
        `add()`, as ONE launch (csrc/envstep.hip: gnbv_rollout_add; same fp32 operation order) -- the observation must
        already sit in its buffer row (in-place env).  GPU tensors only.  `broadcast_first`: every env is bootstrapped
        with terminal_value[0] (the reference's `predict_values(new_obs)[0]`, :206-207)."""
        from .. import _lib
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        assert obs.data_ptr() == self.observations[self.step].data_ptr(), "add_bootstrapped needs the in-place observation row"
        n, t = self.n_envs, self.step
        a = action if (action.dtype == torch.int64 and action.is_contiguous()) else action.to(torch.int64).contiguous()
        es = episode_start if episode_start.dtype in (torch.bool, torch.uint8) else episode_start.bool()
        to = time_outs if time_outs.dtype in (torch.bool, torch.uint8) else time_outs.bool()
        r, tv = reward.reshape(-1), terminal_value.reshape(-1)
        v, lp = value.reshape(-1), log_prob.reshape(-1)
        for x in (r, tv, v, lp):
            assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == n
        _lib.require_cuda(a, r, to, tv, es, v, lp)
        _lib.check(_lib.load().gnbv_rollout_add(
            n, self.actions_shape, a.data_ptr(), r.data_ptr(), to.contiguous().data_ptr(), tv.data_ptr(), 0 if broadcast_first else 1, float(gamma),
            es.contiguous().data_ptr(), v.data_ptr(), lp.data_ptr(), self.actions[t].data_ptr(), self.rewards[t].data_ptr(),
            self.episode_starts[t].data_ptr(), self.values[t].data_ptr(), self.log_probs[t].data_ptr(), _lib.stream_ptr(self.device)),
            "gnbv_rollout_add")
        self.step += 1
        self.pos += 1
        if self.pos == self.buffer_size:
            self.full = True

    def compute_returns_and_advantage(self, last_values: torch.Tensor, dones) -> None:
        if self.rewards.device.type == "cpu":
            # A buffer the caller PLACED on the CPU (BASELINE configs[0]: MLP policy, PPO on the CPU over a recorded feed -- the
            # plain-torch configuration, which never touches the kernels): buffers.py:706-724 as a backward recurrence over the rows,
            #   A_t = d_t + gamma lam m_t A_{t+1},   d_t = r_t + gamma m_t V_{t+1} - V_t,   m_t = 1 - (start of t+1 | done after the last row).
            # Not a fallback: device tensors go to the gfx950 kernel below, which fails loudly without its library.
            t_steps = self.buffer_size
            v_next = last_values.detach().reshape(self.n_envs, 1).to(self.values.dtype)
            alive = 1.0 - torch.as_tensor(dones).reshape(self.n_envs, 1).to(self.values.dtype)
            run = torch.zeros_like(v_next)
            for t in range(t_steps - 1, -1, -1):
                delta = self.rewards[t] + self.gamma * v_next * alive - self.values[t]
                run = delta + self.gamma * self.gae_lambda * alive * run
                self.advantages[t] = run
                v_next = self.values[t]
                alive = 1.0 - self.episode_starts[t].reshape(self.n_envs, 1).to(self.values.dtype)
            self.returns.copy_(self.advantages + self.values)
            return
        gae_ops.compute_returns_and_advantage(self.rewards, self.values, self.episode_starts, last_values.detach(), dones,
                                              self.gamma, self.gae_lambda, advantages=self.advantages, returns=self.returns)

    # ---- minibatches -----------------------------------------------------------
    def rows_of(self, batch_inds: torch.Tensor):
        """flat index i (reference order n*T + t) -> row index t*N + n of the [T, N] layout."""
        t_steps, n = self.buffer_size, self.n_envs
        env = torch.div(batch_inds, t_steps, rounding_mode="floor")
        return (batch_inds - env * t_steps) * n + env

    def get(self, batch_size: Optional[int] = None) -> Generator[RolloutBufferSamples, None, None]:
        assert self.step == self.num_transitions_per_env, ""
        total = self.buffer_size * self.n_envs
        if self._indices_dev is None:
            self._indices_dev = torch.from_numpy(np.asarray(self.indices, dtype=np.int64)).to(self.device)
        if batch_size is None:
            batch_size = total
        start_idx = 0
        while start_idx < total:
            yield self._get_samples(self._indices_dev[int(start_idx):int(start_idx) + int(batch_size)])
            start_idx += batch_size

    def _get_samples(self, batch_inds: torch.Tensor) -> RolloutBufferSamples:
        rows = self.rows_of(batch_inds)
        t, n = self.buffer_size, self.n_envs
        flat = lambda x: x.view(x.shape[0] * n, *x.shape[2:])  # noqa: E731
        if self.lazy_obs:
            from ..ops.encoder_ops import RowGather
            obs = RowGather(flat(self.observations[:t]), rows, None if self.grid_i8 is None else flat(self.grid_i8[:t]),
                            self.compact_state_dim, None if self.autocorr is None else flat(self.autocorr[:t]))
        elif self.compact_state_dim is not None:
            from ..ops.encoder_ops import _flat_rows
            obs = _flat_rows(flat(self.observations[:t])[rows], flat(self.grid_i8[:t])[rows], self.compact_state_dim)
        else:
            obs = flat(self.observations[:t])[rows]
        return RolloutBufferSamples(
            obs, flat(self.actions)[rows], flat(self.values)[rows].flatten(),
            flat(self.log_probs)[rows].flatten(), flat(self.advantages)[rows].flatten(), flat(self.returns)[rows].flatten())

    def flat_values_returns(self):
        """values / returns in the reference's flattened (n*T + t) order, for explained variance."""
        return (self.values.squeeze(-1).transpose(0, 1).reshape(-1), self.returns.squeeze(-1).transpose(0, 1).reshape(-1))
