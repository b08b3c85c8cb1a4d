"""Host side of the PPO minibatch kernels (gennbv_amd/csrc/ppo.hip)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from .. import _lib


class FlatAdam:
    """clip_grad_norm_ + torch.optim.Adam over ONE flat fp32 buffer.

    The module's parameters are re-pointed at slices of `self.params` (and their `.grad` at
    slices of `self.grads`), so autograd accumulates straight into the flat gradient buffer and
    one kernel pair (norm, update) replaces torch's ~60 per-tensor launches
    (stable_baselines3/ppo/ppo_grid_obs.py:271-275; Adam eps 1e-5, policies.py:851-855)."""

    SLOT = 64

    def __init__(self, module: torch.nn.Module, lr: float, betas=(0.9, 0.999), eps: float = 1e-5):
        self.lib = _lib.load()
        ps = [p for p in module.parameters() if p.requires_grad]
        dev = ps[0].device
        _lib.require_cuda(ps[0])
        n = sum(p.numel() for p in ps)
        self.n = n
        self.params = torch.empty(n, dtype=torch.float32, device=dev)
        # one extra slot IN FRONT of the gradient: the rank's approx-KL rides in the same all-reduce as
        # the (small) conv-stack gradients, which come first in parameter order
        # (SLOT floats = 256 bytes, so that the gradient itself stays 256-byte aligned: a 4-byte shift
        # sent every element-wise kernel on the gradients down the unaligned, unvectorised path)
        self.grads_with_slot = torch.zeros(n + self.SLOT, dtype=torch.float32, device=dev)
        self.kl_slot = self.grads_with_slot[:1]
        self.grads = self.grads_with_slot[self.SLOT:]
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.norm_out = torch.zeros(2, dtype=torch.float32, device=dev)
        self.ws = torch.empty(self.lib.gnbv_adam_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.lr, self.betas, self.eps = lr, betas, eps
        off = 0
        self.slices = []
        for p in ps:
            k = p.numel()
            self.params[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.params[off:off + k].view_as(p)
            p.grad = self.grads[off:off + k].view_as(p)
            self.slices.append((off, k))
            off += k
        self.module_params = ps

    def zero_grad(self):
        self.grads.zero_()

    def step(self, max_grad_norm: float, stop_flag: Optional[torch.Tensor] = None, grad_scale: float = 1.0,
             kl_slot_target: Optional[float] = None, rotate=None, sq_slice=None, skip_update: bool = False, loss_finish=None):
        """grad_scale = 1/world and kl_slot_target = target_kl for the data-parallel tail.  rotate = (table [rows, len] int64,
        out [len] int64, counter [1] int32): the launch also leaves the next minibatch's row of `table` in `out`.
        sq_slice = (lo, hi, partial fp64 tensor): sum(grad[lo:hi]^2) was left in `partial` by the kernel that produced that
        gradient slice (gnbv_linear_bwd_dw_sq) -- the norm pass skips the slice; `skip_update`: nor is the slice updated here (its
        update is sharded over the data-parallel ranks: shard_step).  (include/gennbv_hip.h: GnbvAdamStep)"""
        a = _lib.GnbvAdamStep()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq, a.n = (self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                                           self.exp_avg_sq.data_ptr(), self.n)
        a.max_grad_norm = float(max_grad_norm if max_grad_norm is not None else -1.0)
        a.lr, a.beta1, a.beta2, a.eps = float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps)
        a.step, a.stop_flag, a.grad_scale = self.step_count.data_ptr(), _lib.ptr(stop_flag), float(grad_scale)
        a.kl_slot = self.kl_slot.data_ptr() if kl_slot_target is not None else None
        a.target_kl = float(kl_slot_target) if kl_slot_target is not None else -1.0
        a.norm_out, a.workspace, a.workspace_bytes = self.norm_out.data_ptr(), self.ws.data_ptr(), self.ws.numel()
        if rotate is not None:
            table, out, counter = rotate
            assert table.dtype == torch.int64 and table.is_contiguous() and table.dim() == 2 and out.dtype == torch.int64 and out.is_contiguous()
            assert out.numel() == table.shape[1] and counter.dtype == torch.int32
            a.table, a.table_rows, a.row_len, a.out, a.counter = (table.data_ptr(), int(table.shape[0]), int(table.shape[1]), out.data_ptr(),
                                                                   counter.data_ptr())
        if sq_slice is not None:
            lo, hi, part = sq_slice
            assert part.dtype == torch.float64 and part.is_contiguous() and 0 <= lo < hi <= self.n
            a.sq_lo, a.sq_hi, a.sq_partial, a.sq_parts = int(lo), int(hi), part.data_ptr(), int(part.numel())
            if skip_update:
                a.upd_skip_lo, a.upd_skip_hi = int(lo), int(hi)
        if loss_finish is not None:  # a GnbvPpoLoss with defer_stats = 1 (PpoLossOp.args): its statistics are finished inside the norm launch
            a.loss_finish = C.addressof(loss_finish)
        _lib.check(self.lib.gnbv_clip_adam_step_ex(C.byref(a), _lib.stream_ptr(self.params.device)), "gnbv_clip_adam_step_ex")

    # ---- data-parallel replicas: the update of one large slice sharded over the ranks (gennbv_amd/parallel.py) ----
    def enable_shard(self, lo: int, hi: int, rank: int, world: int) -> bool:
        """Rank `rank` of `world` owns parameters [lo + rank * sh, lo + (rank + 1) * sh), sh = (hi - lo) / world, of the slice
        [lo, hi): it receives that shard of the summed gradient (reduce-scatter), keeps Adam moments for it only, updates it and
        all-gathers the slice.  False (nothing changes) when the slice does not divide evenly."""
        if (hi - lo) % world or world < 1:
            return False
        sh = (hi - lo) // world
        dev = self.params.device
        self.shard = {"lo": lo, "hi": hi, "sh": sh, "rank": rank, "world": world,
                      "grad": torch.zeros(sh, dtype=torch.float32, device=dev),
                      "sq": torch.zeros(int(self.lib.gnbv_sq_partials_count()), dtype=torch.float64, device=dev)}
        return True

    def gather_shard_state(self, group=None) -> None:
        """Adam moments of the sharded slice from their owners into every rank's flat buffers (checkpoints: torch_state_dict)."""
        s = getattr(self, "shard", None)
        if s is None or s["world"] <= 1:
            return
        import torch.distributed as dist
        _, m, v = self.shard_views()
        dist.all_gather_into_tensor(self.exp_avg[s["lo"]:s["hi"]], m, group=group)
        dist.all_gather_into_tensor(self.exp_avg_sq[s["lo"]:s["hi"]], v, group=group)

    def shard_sq(self) -> torch.Tensor:
        """sum(shard gradient^2) as fixed-order fp64 partial sums in shard["sq"] (one launch; summed over the ranks by the caller)."""
        s = self.shard
        _lib.check(self.lib.gnbv_sq_partials(s["grad"].data_ptr(), int(s["sh"]), s["sq"].data_ptr(), _lib.stream_ptr(self.params.device)), "gnbv_sq_partials")
        return s["sq"]

    def shard_views(self):
        s = self.shard
        a = s["lo"] + s["rank"] * s["sh"]
        return self.params[a:a + s["sh"]], self.exp_avg[a:a + s["sh"]], self.exp_avg_sq[a:a + s["sh"]]

    def shard_step(self, stop_flag: Optional[torch.Tensor] = None):
        """Adam on this rank's shard from `shard["grad"]` with the clip factor the main step() of this optimizer step computed."""
        p, m, v = self.shard_views()
        _lib.check(self.lib.gnbv_adam_shard_step(p.data_ptr(), self.shard["grad"].data_ptr(), m.data_ptr(), v.data_ptr(), int(self.shard["sh"]),
                                                 self.norm_out.data_ptr(), float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                                 self.step_count.data_ptr(), _lib.ptr(stop_flag), _lib.stream_ptr(self.params.device)),
                   "gnbv_adam_shard_step")

    def slice_of(self, param: torch.Tensor):
        """(lo, hi) of a parameter in the flat buffers, or None."""
        for (off, k), p in zip(self.slices, self.module_params):
            if p is param:
                return off, off + k
        return None

    def torch_state_dict(self, template: torch.optim.Adam) -> dict:
        """This optimizer's state in torch.optim.Adam.state_dict() format (checkpoints: policy.optimizer.pth);
        `template` = the policy's (idle) torch Adam, which supplies the param_groups."""
        sd = template.state_dict()
        step = int(self.step_count.item())
        sd["state"] = {}
        if step > 0:
            for i, ((off, k), p) in enumerate(zip(self.slices, self.module_params)):
                sd["state"][i] = {"step": torch.tensor(float(step)), "exp_avg": self.exp_avg[off:off + k].view_as(p).clone(),
                                  "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p).clone()}
        for g in sd["param_groups"]:
            g["lr"] = float(self.lr)
        return sd

    def load_torch_state_dict(self, sd: dict) -> None:
        """Adopt exp_avg / exp_avg_sq / step from a torch.optim.Adam.state_dict() over the same parameters."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count.zero_()
        for i, (off, k) in enumerate(self.slices):
            st = sd["state"].get(i)
            if st:
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count.fill_(int(st["step"]))

    def load_torch_adam_state(self, opt: torch.optim.Adam):
        """Adopt exp_avg / exp_avg_sq / step of a torch Adam over the same parameters (checkpoints)."""
        for (off, k), p in zip(self.slices, self.module_params):
            st = opt.state.get(p)
            if st:
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count.fill_(int(st["step"]))


class PpoLossOp:
    """Static buffers + the fused loss/gradient kernel for minibatches of a fixed size."""

    def __init__(self, batch: int, head_dims: List[int], device, max_rows: int, clip_range: float, clip_range_vf,
                 ent_coef: float, vf_coef: float, policy_scale: float, target_kl, normalize_advantage: bool = True):
        self.lib = _lib.load()
        self.batch, self.head_dims = batch, list(head_dims)
        n_logits, nh = sum(head_dims), len(head_dims)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)  # noqa: E731
        # [row numbers (batch) | (mean, 1 / (std + 1e-8)) of the minibatch's advantages as two fp32 in one int64 slot]: ONE buffer, so
        # that the row rotation of the replayed graph (FlatAdam.step(rotate=...)) delivers both
        # ... | the sum of the minibatch's input-autocorrelation rows, 768 int32 (GnbvEncoderParams.autocorr_total)]
        self.rows_ext = z(batch + 1 + 384, dt=torch.int64)
        self.rows = self.rows_ext[:batch]
        self.adv_slot = self.rows_ext[batch:batch + 1].view(torch.float32)
        self.ac_slot = self.rows_ext[batch + 1:].view(torch.int32)
        self.actions, self.old_values, self.old_log_prob = z(batch, nh), z(batch), z(batch)
        self.advantages, self.returns = z(batch), z(batch)
        self.d_logits, self.d_values = z(batch, n_logits), z(batch)
        self.stats = z(max_rows + 1, 8)
        self.stats_row = z(1, dt=torch.int64)
        self.stop_flag = z(1, dt=torch.int32)
        a = _lib.GnbvPpoLoss()
        a.batch, a.n_logits, a.n_heads = batch, n_logits, nh
        for i, d in enumerate(head_dims):
            a.head_dims[i] = int(d)
        a.normalize_advantage = int(normalize_advantage)
        a.clip_range = float(clip_range)
        a.clip_range_vf = float(clip_range_vf) if clip_range_vf is not None else -1.0
        a.ent_coef, a.vf_coef, a.policy_scale = float(ent_coef), float(vf_coef), float(policy_scale)
        a.target_kl = float(target_kl) if target_kl is not None else -1.0
        a.actions, a.old_values, a.old_log_prob = self.actions.data_ptr(), self.old_values.data_ptr(), self.old_log_prob.data_ptr()
        a.advantages, a.returns = self.advantages.data_ptr(), self.returns.data_ptr()
        a.d_logits, a.d_values = self.d_logits.data_ptr(), self.d_values.data_ptr()
        a.head_entropy, a.head_lse, a.kl_out, a.rows = None, None, None, None
        a.stats, a.stats_row, a.stop_flag = self.stats.data_ptr(), self.stats_row.data_ptr(), self.stop_flag.data_ptr()
        self.scratch = z(8 * batch + 64)  # per-sample terms + the completion counter (zero-initialised once)
        a.adv_norm = None
        a.defer_stats = 0
        a.scratch = self.scratch.data_ptr()
        self.args = a
        self.device = device

    def finish_stats(self) -> None:
        """The statistics row / KL slot of the last call with `args.defer_stats = 1`, as a launch of its own on the current stream
        (gnbv_ppo_loss_finish) -- for callers whose optimizer launch cannot carry it (GnbvAdamStep.loss_finish): the data-parallel step needs
        the rank's KL in front of the gradient exchange."""
        _lib.check(_lib.load().gnbv_ppo_loss_finish(C.byref(self.args), _lib.stream_ptr(self.device)), "gnbv_ppo_loss_finish")

    def bind(self, buf):
        """Fused gather: the loss kernel reads actions / values / log_probs / advantages / returns of rows
        `self.rows` (row = t*N + n) straight out of the rollout buffer (GnbvPpoLoss.rows)."""
        a = self.args
        for t in (buf.actions, buf.values, buf.log_probs, buf.advantages, buf.returns):
            assert t.is_contiguous() and t.dtype == torch.float32
        a.actions, a.old_values, a.old_log_prob = buf.actions.data_ptr(), buf.values.data_ptr(), buf.log_probs.data_ptr()
        a.advantages, a.returns = buf.advantages.data_ptr(), buf.returns.data_ptr()
        a.rows = self.rows.data_ptr()

    def gather(self, buf):
        """actions / values / log_probs / advantages / returns of rows `self.rows` (row = t*N + n)."""
        a = self.args
        a.actions, a.old_values, a.old_log_prob = self.actions.data_ptr(), self.old_values.data_ptr(), self.old_log_prob.data_ptr()
        a.advantages, a.returns = self.advantages.data_ptr(), self.returns.data_ptr()
        a.rows = None
        _lib.check(self.lib.gnbv_gather_minibatch(
            self.rows.data_ptr(), self.batch, self.actions.shape[1], buf.actions.data_ptr(), buf.values.data_ptr(),
            buf.log_probs.data_ptr(), buf.advantages.data_ptr(), buf.returns.data_ptr(), self.actions.data_ptr(),
            self.old_values.data_ptr(), self.old_log_prob.data_ptr(), self.advantages.data_ptr(), self.returns.data_ptr(),
            _lib.stream_ptr(self.device)), "gnbv_gather_minibatch")

    def __call__(self, logits: torch.Tensor, values: torch.Tensor):
        assert logits.is_contiguous() and values.is_contiguous() and logits.dtype == torch.float32
        self.args.logits, self.args.values = logits.data_ptr(), values.data_ptr()
        _lib.check(self.lib.gnbv_ppo_loss(C.byref(self.args), _lib.stream_ptr(self.device)), "gnbv_ppo_loss")
        return self.d_logits, self.d_values
