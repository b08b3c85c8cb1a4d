"""Oracle replay of sampled environments over a finished rollout (TEST INFRASTRUCTURE: the checker, never the product).

`check_rollout(algo, envs)` takes a PPO_Grid_Obs whose rollout buffer holds the last `collect_rollouts()` of a
ReplayFeedEnv and, for each sampled env, replays the tail of that rollout -- from the first episode boundary inside
the buffer to its last row -- through the CPU oracle (oracle/env_oracle.py + oracle/oracle.c: the restatement of
env_train_gennbv.py:246-457 pinned by F5), fed with the very frames and the very actions the timed kernels consumed,
and compares bit for bit:

  * every stored observation row (pose history, tri-class grid -- int8 rows or the flat fp32 slice --, gray frames),
  * rewards (steps without an episode end: the buffer adds the time-out bootstrap to the others) and episode starts,
  * the env's probability grid and scanned-GT grid after the last step.

Used by tests/test_fullsize_gpu.py and, after the timed region, by bench.py (`timed_state_check`)."""
from __future__ import annotations

import numpy as np
import torch


def _oracle_env_after_done(env, es):
    """An OracleEnv over the envs `es` in the state reset_idx (env_train_gennbv.py:377-436) leaves behind."""
    from oracle.env_oracle import OracleEnv
    upd = env.updater
    es = torch.as_tensor(list(es), device=upd.range_gt.device)
    o = OracleEnv(env.cfg, upd.inv_intri_host.numpy(), upd.range_gt[es].cpu().numpy(), upd.voxel_size_gt[es].cpu().numpy(),
                  upd.grid_gt[es].cpu().numpy(), env.num_valid_voxel_gt[es].cpu().numpy(),
                  max_episode_length=env.max_episode_length)
    o.pending_reset[:] = 1  # grids are zeroed before the next update; histories / counters are already at their reset values
    return o


def check_rollout(algo, envs, cursor_end=None, max_steps=None, across_ends=0):
    """-> dict(status="bit-exact" | first mismatch, envs=[...], steps_replayed=int, episode_ends_seen=int, ...).
    `cursor_end`: the feed cursor right after the rollout (default: now).  `max_steps`: replay at most this many trailing env steps per
    env.  `across_ends`: additionally replay up to this many envs (the first ones found, beside `envs`) that END an episode inside the
    buffer after their first boundary, from that first boundary to the last row whatever `max_steps` says -- the done -> deferred
    zeroing -> next-update hand-over (env_train_gennbv.py:377-436) at the checked size.  Envs that start at the same buffer row are
    replayed as ONE multi-env oracle (OpenMP over its envs)."""
    env, buf = algo.env, algo.rollout_buffer
    cfg, feed = env.cfg, env.feed
    t_steps, n = buf.buffer_size, buf.n_envs
    assert buf.step == t_steps, "the rollout buffer must hold a finished rollout"
    cursor_end = feed.cursor if cursor_end is None else cursor_end
    nf = feed.num_frames
    s0, ge = cfg.state_dim, cfg.grid_dim
    compact = buf.compact_state_dim is not None
    starts = buf.episode_starts[:, :, 0].cpu().numpy().astype(bool)  # [T, N]
    last_starts = algo._last_episode_starts.cpu().numpy().astype(bool)
    prob_all, scan_all = env.prob_grid, env.scanned_gt_grid
    out = {"status": "bit-exact", "envs": [], "steps_replayed": 0, "episode_ends_seen": 0,
           "compared": "obs rows (pose history, tri-class grid, gray frames), rewards, episode starts, final prob / scanned grids"}

    def fail(msg):
        out["status"] = "MISMATCH: " + msg
        return out

    groups = {}  # first replayed row -> envs
    envs = [int(e) for e in envs]
    if across_ends:
        ends_after = lambda e, t0: bool(starts[t0 + 1:, e].any()) or bool(last_starts[e])  # noqa: E731
        extra = []
        for e in range(n):
            b = np.flatnonzero(starts[1:, e]) + 1
            if e not in envs and len(b) and ends_after(e, int(b[0])):
                extra.append(e)
                groups.setdefault(int(b[0]), []).append(e)
                if len(extra) >= across_ends:
                    break
        out["across_ends_envs"] = extra
    for e in envs:
        cand = [t for t in range(1, t_steps) if starts[t, e]]
        if max_steps is not None:
            late = [t for t in cand if t >= t_steps - max_steps]
            cand = late[:1] if late else cand[-1:]
        if not cand:
            out["envs"].append({"env": e, "skipped": "no episode boundary inside the buffer"})
            continue
        groups.setdefault(cand[0], []).append(e)

    for t0, es in sorted(groups.items()):
        k = len(es)
        es_t = torch.as_tensor(es, device=buf.actions.device)
        orc_env = _oracle_env_after_done(env, es)
        acts = buf.actions[t0:][:, es_t].cpu().numpy().astype(np.int64)           # [T', k, 6]
        rews = buf.rewards[t0:][:, es_t, 0].cpu().numpy()                          # [T', k]
        # The buffer's rewards carry the time-out bootstrap gamma * V * infos["time_outs"] (on_policy_algorithm_grid_obs.py:205-208),
        # and infos["time_outs"] is only refreshed on steps in which SOME env (of all n) resets (reference quirk, env_oracle.py header):
        # an env's entry is known from the first such step on; rewards are compared wherever it is known to be False.
        stale_time_out = [None] * k
        fe = feed.depth_raw.device
        es_f = es_t.to(fe)
        for t in range(t0, t_steps):
            f = (cursor_end - (t_steps - t)) % nf
            rgba = (feed.rgba[f][es_f].cpu().numpy() if feed.rgba is not None
                    else np.zeros((k, cfg.camera_height, cfg.camera_width, 4), np.uint8))
            obs, rew, done, info = orc_env.step(acts[t - t0], feed.depth_raw[f][es_f].cpu().numpy(), feed.seg_raw[f][es_f].cpu().numpy(),
                                                rgba, feed.c2w[f][es_f].cpu().numpy())
            rows_small = buf.observations[t + 1][es_t].cpu().numpy()
            rows_grid = buf.grid_i8[t + 1][es_t].cpu().numpy() if compact else None
            any_reset = bool(starts[t + 1].any()) if t + 1 < t_steps else bool(last_starts.any())
            for j, e in enumerate(es):
                row = rows_small[j]
                if compact:
                    if row[:s0].tobytes() != obs[j, :s0].tobytes():
                        return fail(f"env {e} step {t}: pose history")
                    if row[s0:].tobytes() != obs[j, s0 + ge:].tobytes():
                        return fail(f"env {e} step {t}: gray frames")
                    if not np.array_equal(rows_grid[j], obs[j, s0:s0 + ge].astype(np.int8)):
                        return fail(f"env {e} step {t}: tri-class grid (int8 row)")
                elif row.tobytes() != obs[j].tobytes():
                    return fail(f"env {e} step {t}: flat observation row")
                if any_reset:
                    stale_time_out[j] = bool(done[j]) and bool(info["time_outs"][j])
                if stale_time_out[j] is False:
                    if rews[t - t0, j].tobytes() != rew[j].tobytes():
                        return fail(f"env {e} step {t}: reward {rews[t - t0, j]!r} vs oracle {rew[j]!r}")
                    out["rewards_compared"] = out.get("rewards_compared", 0) + 1
                nxt = starts[t + 1, e] if t + 1 < t_steps else last_starts[e]
                if bool(nxt) != bool(done[j]):
                    return fail(f"env {e} step {t}: done flag")
                # an episode end FOLLOWED by replayed steps (the last row's done has no update behind it inside this rollout)
                out["episode_ends_seen"] += int(done[j]) if t + 1 < t_steps else 0
                out["steps_replayed"] += 1
        # grids as the env holds them now: zeroing after a done is deferred to the next update on both sides
        for j, e in enumerate(es):
            if prob_all[e].cpu().numpy().tobytes() != orc_env.prob_grid[j].tobytes():
                return fail(f"env {e}: probability grid after the last step")
            if scan_all[e].cpu().numpy().tobytes() != orc_env.scanned_gt_grid[j].tobytes():
                return fail(f"env {e}: scanned-GT grid after the last step")
            out["envs"].append({"env": e, "from_step": t0, "steps": t_steps - t0})
    if not out["steps_replayed"]:
        out["status"] = "not checked (no sampled env had an episode boundary inside the buffer)"
    return out
