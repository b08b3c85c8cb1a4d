"""The policy evaluation of a rollout step (on_policy_algorithm_grid_obs.py:160-168 of the reference: `self.policy(obs)`) with everything
that does not change between two env steps hoisted out of the step.

`ActorCriticPolicy_Train_Eval.forward` on the gfx950 kernels is six library calls (csrc/encoder.hip, linear.hip, head.hip) plus a fork /
join of the pose branch onto the second stream.  Through the general path (ops/encoder_ops.py: hybrid_branches -> grid_encoder ->
linear_relu -> policy_head) every call re-derives what it needs from the modules -- shapes, workspaces, the parameter struct, ~15
intermediate tensors -- ~280 us of host time per env step, more than the GPU needs for the kernels of the whole step on the pool's slower
hosts (tools/profile_rollout_host.py).  During collect_rollouts the parameters are FIXED (they change in train() only), so:

* `RolloutForward.build(policy, n)`: checks that the policy takes exactly the kernels below (else None: the general path runs), looks
  the modules up once, allocates the intermediates once;
* `prepare()`, once per collect_rollouts: the parameter-only launches of the inference forward -- BatchNorm scale / shift of both layers
  from the running statistics, the conv2 weight images (gnbv_encoder_eval_prepare, GnbvEncoderParams.eval_prepared) -- which the general
  path repeats in every step, three small kernels on the step's critical path;
* `__call__(dense_obs)`: the six calls with precomputed arguments -> (logits [n, A], values [n]), bit-identical to the general path
  (tests/test_rollout_gpu.py).

Two forms, chosen by what the general path would do for the same policy and observation (so that the plan stays bit-identical to it):
* "compact" (the bench's kernel set, G = 64): compact int8 rows -> the one-launch inference conv kernel -> fc_grid with BatchNorm-2 folded into
  its operand load;
* "flat" (round 6; the reference's own 20^3 workload, where fc_grid cannot fold): fp32 observation rows -> the fp32 conv kernels + BatchNorm
  launches of gnbv_encoder_grid_forward -> materialised features -> fc_grid.  No parameter-only launches to hoist there (prepare() only looks
  the arguments up); what the plan removes is the general path's host time -- at 20^3 the rollout step is HOST-bound: 185 us of kernels in 375.

Same kernels, same arguments, same streams as ops/encoder_ops.py -- this file adds no arithmetic."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import _lib
from . import encoder_ops


class RolloutForward:
    def __init__(self):
        raise TypeError("use RolloutForward.build")

    @classmethod
    def build(cls, policy, n: int) -> Optional["RolloutForward"]:
        """None when the policy's inference forward is not exactly: compact int8 rows -> one-launch conv1 + conv2 -> fc_grid with
        BatchNorm-2 folded into its operand load | pose encode -> two split-K linears on the second stream -> fused policy head."""
        enc = policy.features_extractor
        lin_g = getattr(enc, "output_layer_grid", [None])[0]
        seq_a = getattr(enc, "naive_encoder_action", None)
        ok = (getattr(policy, "_fused_rollout", False) and getattr(enc, "backend", "") == "hip" and not getattr(enc, "semantic_branch", False)
              and getattr(enc, "overlap_branches", False) and not getattr(enc, "force_fp32", False) and isinstance(lin_g, torch.nn.Linear)
              and seq_a is not None and len(seq_a) == 4 and isinstance(seq_a[0], torch.nn.Linear) and isinstance(seq_a[2], torch.nn.Linear)
              and enc.state_input_shape[0] % 6 == 0
              and encoder_ops.policy_head_supported(enc, policy.action_net, policy.value_net))
        if not ok:
            return None
        lins = (seq_a[0], seq_a[2], lin_g)
        for lin in lins:
            nn_, k = lin.weight.shape
            if (getattr(lin, "_fp32_arith", False) or k % 8 or k < 64 or nn_ % 64 or not lin.weight.is_contiguous() or lin.weight.dtype != torch.float32):
                return None
        self = object.__new__(cls)
        lib = self.lib = _lib.load()
        self.policy, self.enc, self.n = policy, enc, int(n)
        dev = self.dev = lin_g.weight.device
        s, g = enc.state_input_shape[0], enc.grid_size
        self.s, self.g = s, g
        o2 = encoder_ops.conv_out(encoder_ops.conv_out(g))
        self.p2 = o2 ** 3
        # (what the general path decides per call from the same arguments: hybrid_branches -> linear_fold_ok)
        self.fold = bool(encoder_ops.linear_fold_ok(lin_g, n, self.p2, False))
        self.seq = enc.naive_encoder_grid
        self.lins = lins
        f32 = dict(dtype=torch.float32, device=dev)
        # intermediates: consumed inside the call that writes them (stream-ordered), so one set serves every step.  y1 is not written by
        # the one-launch inference kernel (csrc/encoder.hip fused_eval) but the entry point wants a valid pointer: the general path's size
        self.y1 = torch.empty(lib.gnbv_encoder_y1_elems(n, g), **f32)
        self.y2 = torch.empty(n * 16 * self.p2, **f32)
        self.bn_state = torch.empty(2 * 4 * 16 + 768, **f32)
        self.feats = None if self.fold else torch.empty(n, 16 * self.p2, **f32)  # ("flat": BatchNorm-2 + ReLU materialised for fc_grid)
        self.ws_enc = encoder_ops._workspace(lib, n, g, dev)
        self.pose_in = torch.empty(n, 4 * s, **f32)
        self.h1 = torch.empty(n, seq_a[0].out_features, **f32)
        self.fa = torch.empty(n, seq_a[2].out_features, **f32)
        self.fg = torch.empty(n, lin_g.out_features, **f32)
        self.feat = torch.empty(n, enc.output_layer[0].out_features, **f32)
        self.n_act = policy.action_net.out_features
        # the split-K workspaces of the three linears: the SAME cache entries the general path uses (one per layer, keyed by its weight)
        self.ws_lin = []
        for lin, m_k in zip(lins, (4 * s, seq_a[0].out_features, 16 * self.p2)):
            key = ("lin", n, lin.out_features, m_k, str(dev), lin.weight.data_ptr())
            ws = encoder_ops._ws_cache.get(key)
            if ws is None:
                ws = torch.empty(lib.gnbv_linear_workspace_bytes(n, lin.out_features, m_k), dtype=torch.uint8, device=dev)
                encoder_ops._ws_cache[key] = ws
            self.ws_lin.append(ws)
        self.side = encoder_ops._side_stream(dev)
        self.ev_fork, self.ev_join = torch.cuda.Event(), torch.cuda.Event()
        self.params = None
        self.prepared = False
        self._sig = None
        return self

    # ------------------------------------------------------------------------------------------------------------------
    def _signature(self):
        """what the cached arguments were derived from: a re-parametrised / re-allocated policy must rebuild the plan"""
        enc = self.enc
        # (+ the tensors' in-place version counters and the BatchNorm running statistics the prepared scale / shift came from: a
        # load_state_dict / set_parameters from a callback in the MIDDLE of a rollout sends the rest of it through the general path)
        # Evaluated before EVERY policy evaluation: the tensors are fetched from the modules' own dicts (`_sig_slots`: whatever tensor a
        # slot holds NOW is what is compared), not through nn.Module.__getattr__ / Sequential.__getitem__ -- that form cost 47 us per env
        # step of a rollout that is host-bound at 20^3 (tools/profile_rollout_host.py).
        slots = self.__dict__.get("_sig_slots")
        if slots is None:
            pol = self.policy
            mods = (self.seq[0], self.seq[1], self.seq[3], self.seq[4], *self.lins, enc.output_layer[0], pol.action_net, pol.value_net)
            slots = [(m._parameters, "weight") for m in mods]
            for m in (self.seq[1], self.seq[4]):
                slots += [(m._parameters, "weight"), (m._parameters, "bias"), (m._buffers, "running_mean"), (m._buffers, "running_var")]
            self._sig_slots = slots
        ts = [d[k] for d, k in slots]
        return (tuple((t.data_ptr(), t._version) for t in ts), bool(enc.training), bool(enc.__dict__.get("force_fp32", False)),
                tuple(bool(lin.__dict__.get("_fp32_arith", False)) for lin in self.lins), id(enc._buffers.get("_range_flag")))

    def prepare(self) -> bool:
        """Once per collect_rollouts, before its first policy evaluation.  False: this rollout runs through the general path."""
        enc = self.enc
        self.prepared = False
        if enc.training or getattr(enc, "force_fp32", False) or any(getattr(lin, "_fp32_arith", False) for lin in self.lins):
            return False
        flag = getattr(enc, "_range_flag", None)
        self.flag_ptr = None if flag is None else flag.data_ptr()
        # (grid_i8 pointer / stride are filled in per call; a dummy row makes the path predicates of the prepare call meaningful)
        self.params = encoder_ops._params_struct(self.seq, None, None, None, (False, flag, None))
        if self.fold:
            self.params.grid_i8 = self.y1.data_ptr()  # any 16-byte aligned device pointer: the prepare launches do not read it
            self.params.grid_i8_row_stride = self.g ** 3
            st = _lib.stream_ptr(self.dev)
            err = self.lib.gnbv_encoder_eval_prepare(self.n, self.g, C.byref(self.params), self.bn_state.data_ptr(), self.ws_enc.data_ptr(),
                                                     self.ws_enc.numel(), st)
            if err == -2:  # GNBV_ERR_NOT_APPLICABLE: no one-launch inference kernel for this (parameters, grid)
                return False
            _lib.check(err, "gnbv_encoder_eval_prepare")
            self.params.eval_prepared = 1
        # ("flat": the conv forward issues its BatchNorm launches itself, every step, as on the general path)
        self._sig = self._signature()
        pol, lo = self.policy, enc.output_layer[0]
        self.head_w = (lo.weight.data_ptr(), lo.bias.data_ptr(), lo.out_features, pol.action_net.weight.data_ptr(), pol.action_net.bias.data_ptr(),
                       self.n_act, pol.value_net.weight.data_ptr(), pol.value_net.bias.data_ptr())
        self.lin_w = [(lin.weight.data_ptr(), lin.bias.data_ptr(), lin.out_features, lin.in_features) for lin in self.lins]
        self.prepared = True
        return True

    def applies_to(self, obs) -> bool:
        if not self.prepared or torch.is_grad_enabled():
            return False
        if self.fold:  # "compact"
            return (isinstance(obs, encoder_ops.DenseObs) and obs.compact_state_dim is not None and obs.base.shape[0] == self.n
                    and obs.base.dtype == torch.float32 and obs.grid_i8 is not None and obs.grid_i8.stride(0) % 16 == 0
                    and obs.grid_i8.data_ptr() % 16 == 0 and self._sig == self._signature())
        # "flat": plain fp32 observation rows [n, state | G^3 | ...] (no int8 side copy: that is another kernel set on the general path)
        return (isinstance(obs, torch.Tensor) and obs.dim() == 2 and obs.shape[0] == self.n and obs.dtype == torch.float32 and obs.is_cuda
                and obs.stride(1) == 1 and obs.shape[1] >= self.s + self.g ** 3 and obs.data_ptr() % 16 == 0 and self._sig == self._signature())

    def __call__(self, obs, tail=None):
        """(logits [n, A], values [n]) of the compact observation rows `obs` (encoder_ops.DenseObs).  `tail(raw_stream)`: more work for
        the second stream, issued behind the pose branch and NOT joined here (the caller joins `self.side` when it needs the result)."""
        lib, n, dev = self.lib, self.n, self.dev
        st = _lib.stream_ptr(dev)
        cur = torch.cuda.current_stream(dev)
        self.ev_fork.record(cur)  # the fork point; the pose kernels are issued after the grid branch (encoder_ops.hybrid_branches)
        p = self.params
        w, b, nn_, k = self.lin_w[2]
        ws = self.ws_lin[2]
        if self.fold:
            base, g8 = obs.base, obs.grid_i8
            p.grid_i8, p.grid_i8_row_stride = g8.data_ptr(), int(g8.stride(0))
            _lib.check(lib.gnbv_encoder_grid_forward(None, None, base.stride(0), n, self.g, C.byref(p), 0, None, self.y1.data_ptr(), self.y2.data_ptr(),
                                                     self.bn_state.data_ptr(), None, self.ws_enc.data_ptr(), self.ws_enc.numel(), st),
                       "gnbv_encoder_grid_forward")
            sc = self.bn_state.data_ptr() + 4 * 64
            _lib.check(lib.gnbv_linear_forward_fold(self.y2.data_ptr(), sc, sc + 4 * 16, self.p2, self.flag_ptr, w, b, n, nn_, k, 1, self.fg.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), st), "gnbv_linear_forward_fold")
        else:
            base = obs
            _lib.check(lib.gnbv_encoder_grid_forward(base.data_ptr() + 4 * self.s, None, base.stride(0), n, self.g, C.byref(p), 0, None, self.y1.data_ptr(),
                                                     self.y2.data_ptr(), self.bn_state.data_ptr(), self.feats.data_ptr(), self.ws_enc.data_ptr(),
                                                     self.ws_enc.numel(), st), "gnbv_encoder_grid_forward")
            _lib.check(lib.gnbv_linear_forward(self.feats.data_ptr(), w, b, n, nn_, k, 1, self.fg.data_ptr(), ws.data_ptr(), ws.numel(), st),
                       "gnbv_linear_forward")
        side = self.side
        side.wait_event(self.ev_fork)
        sst = side.cuda_stream
        _lib.check(lib.gnbv_pose_encode(base.data_ptr(), None, base.stride(0), n, self.s // 6, self.pose_in.data_ptr(), sst), "gnbv_pose_encode")
        x = self.pose_in
        for i, out in ((0, self.h1), (1, self.fa)):
            w, b, nn_, k = self.lin_w[i]
            ws = self.ws_lin[i]
            _lib.check(lib.gnbv_linear_forward(x.data_ptr(), w, b, n, nn_, k, 1, out.data_ptr(), ws.data_ptr(), ws.numel(), sst), "gnbv_linear_forward")
            x = out
        self.ev_join.record(side)
        cur.wait_event(self.ev_join)
        if tail is not None:  # (behind the pose branch: in front of it the step measured 10-13 us slower -- the join waits for the branch)
            tail(sst)
        logits = torch.empty(n, self.n_act, dtype=torch.float32, device=dev)  # (kept by the caller across the next step: not plan-owned)
        values = torch.empty(n, dtype=torch.float32, device=dev)
        w_out, b_out, f, w_act, b_act, a, w_val, b_val = self.head_w
        _lib.check(lib.gnbv_policy_head_forward(self.fa.data_ptr(), self.fg.data_ptr(), n, self.fa.shape[1], self.fg.shape[1], w_out, b_out, f, w_act, b_act,
                                                a, w_val, b_val, self.feat.data_ptr(), logits.data_ptr(), values.data_ptr(), st), "gnbv_policy_head_forward")
        return logits, values
