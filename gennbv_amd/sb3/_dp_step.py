"""The data-parallel optimizer step of PPO_Grid_Obs (gennbv_amd/parallel.py): phase A -> exchange of the late
gradients issued from the second stream, overlapped with the conv backward -> all-reduce of the conv gradients + KL slot -> clip / Adam
tail, sharded or replicated update.  The reference has no multi-GPU path.  (Split out of ppo_grid_obs.py in round 6.)"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F


class DataParallelStepMixin:
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
