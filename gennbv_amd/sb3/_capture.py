"""hipGraph capture of one PPO minibatch (single-GPU and RCCL data-parallel) and the choice between capture candidates.
(Split out of ppo_grid_obs.py in round 6.)"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F


class GraphCaptureMixin:
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
            k = int(os.environ.get("GENNBV_GRAPH_CANDIDATES", "3")) if (st.get("replays_per_call", 0) >= 256 and stable) else 1
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
