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


def _oracle_env_after_done(env, e: int):
    """A one-env OracleEnv in the state reset_idx (env_train_gennbv.py:377-436) leaves behind."""
    from oracle.env_oracle import OracleEnv
    upd = env.updater
    o = OracleEnv(env.cfg, upd.inv_intri_host.numpy(), upd.range_gt[e:e + 1].cpu().numpy(), upd.voxel_size_gt[e:e + 1].cpu().numpy(),
                  upd.grid_gt[e:e + 1].cpu().numpy(), env.num_valid_voxel_gt[e:e + 1].cpu().numpy(),
                  max_episode_length=env.max_episode_length)
    o.pending_reset[:] = 1  # grids are zeroed before the next update; histories / counters are already at their reset values
    return o


def check_rollout(algo, envs, cursor_end=None, max_steps=None):
    """-> dict(status="bit-exact" | first mismatch, envs=[...], steps_replayed=int, ...).  `cursor_end`: the feed cursor
    right after the rollout (default: now).  `max_steps`: replay at most this many trailing env steps per env."""
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

    for e in envs:
        e = int(e)
        cand = [t for t in range(1, t_steps) if starts[t, e]]
        if max_steps is not None:
            late = [t for t in cand if t >= t_steps - max_steps]
            cand = late[:1] if late else cand[-1:]
        if not cand:
            out["envs"].append({"env": e, "skipped": "no episode boundary inside the buffer"})
            continue
        t0 = cand[0]
        orc_env = _oracle_env_after_done(env, e)
        acts = buf.actions[t0:, e].cpu().numpy().astype(np.int64)
        rews = buf.rewards[t0:, e, 0].cpu().numpy()
        rows_small = buf.observations[t0 + 1:, e].cpu().numpy()
        rows_grid = buf.grid_i8[t0 + 1:, e].cpu().numpy() if compact else None
        # The buffer's rewards carry the time-out bootstrap gamma * V * infos["time_outs"] (on_policy_algorithm_grid_obs.py:205-208),
        # and infos["time_outs"] is only refreshed on steps in which SOME env resets (reference quirk, env_oracle.py header): this
        # env's entry is known from the first such step on; rewards are compared wherever it is known to be False.
        stale_time_out = None
        for t in range(t0, t_steps):
            f = (cursor_end - (t_steps - t)) % nf
            rgba = feed.rgba[f, e:e + 1].cpu().numpy() if feed.rgba is not None else np.zeros((1, cfg.camera_height, cfg.camera_width, 4), np.uint8)
            obs, rew, done, info = orc_env.step(acts[t - t0:t - t0 + 1], feed.depth_raw[f, e:e + 1].cpu().numpy(), feed.seg_raw[f, e:e + 1].cpu().numpy(),
                                                rgba, feed.c2w[f, e:e + 1].cpu().numpy())
            row = rows_small[t - t0]
            if compact:
                if row[:s0].tobytes() != obs[0, :s0].tobytes():
                    return fail(f"env {e} step {t}: pose history")
                if row[s0:].tobytes() != obs[0, s0 + ge:].tobytes():
                    return fail(f"env {e} step {t}: gray frames")
                if not np.array_equal(rows_grid[t - t0], obs[0, s0:s0 + ge].astype(np.int8)):
                    return fail(f"env {e} step {t}: tri-class grid (int8 row)")
            elif row.tobytes() != obs[0].tobytes():
                return fail(f"env {e} step {t}: flat observation row")
            any_reset = bool(starts[t + 1].any()) if t + 1 < t_steps else bool(last_starts.any())
            if any_reset:
                stale_time_out = bool(done[0]) and bool(info["time_outs"][0])
            if stale_time_out is False:
                if rews[t - t0].tobytes() != rew[0].tobytes():
                    return fail(f"env {e} step {t}: reward {rews[t - t0]!r} vs oracle {rew[0]!r}")
                out["rewards_compared"] = out.get("rewards_compared", 0) + 1
            nxt = starts[t + 1, e] if t + 1 < t_steps else last_starts[e]
            if bool(nxt) != bool(done[0]):
                return fail(f"env {e} step {t}: done flag")
            out["episode_ends_seen"] += int(done[0])
            out["steps_replayed"] += 1
        # grids as the env holds them now: zeroing after a done is deferred to the next update on both sides
        if prob_all[e].cpu().numpy().tobytes() != orc_env.prob_grid[0].tobytes():
            return fail(f"env {e}: probability grid after the last step")
        if scan_all[e].cpu().numpy().tobytes() != orc_env.scanned_gt_grid[0].tobytes():
            return fail(f"env {e}: scanned-GT grid after the last step")
        out["envs"].append({"env": e, "from_step": t0, "steps": t_steps - t0})
    if not out["steps_replayed"]:
        out["status"] = "not checked (no sampled env had an episode boundary inside the buffer)"
    return out
