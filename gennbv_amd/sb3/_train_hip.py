"""PPO_Grid_Obs.train() on the gfx950 kernels: set-up of the fused update (flat Adam, fused loss, write-through gradients), the minibatch
body (gather -> forward -> loss -> backward -> clip + Adam, no host synchronisation), the per-call tables and the epochs x minibatches
loop, the range guard that repeats a flagged call on the fp32-MFMA kernels.  Reference: stable_baselines3/ppo/ppo_grid_obs.py:176-297.
(Split out of ppo_grid_obs.py in round 6; PPO_Grid_Obs inherits these methods.)"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F


class FusedTrainMixin:
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

    def _check_ranges(self) -> bool:
        """Hybrid_Encoder.check_operand_ranges (split-f16 kernels' limits made loud) -> whether the encoder is on the fp32 kernels."""
        enc = self.policy.features_extractor
        if getattr(enc, "backend", "") != "hip" or not hasattr(enc, "check_operand_ranges"):
            return False
        return bool(enc.check_operand_ranges()["force_fp32"])
