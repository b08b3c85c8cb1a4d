"""Timeline of ONE PPO minibatch step out of a rocprofv3 kernel-trace database: every kernel between two
consecutive launches of a marker kernel (default k_ppo_logp: once per minibatch), with start offset, duration and queue/stream id, plus the
per-queue busy time and the union (critical-path) time.

    python tools/rocprof_timeline.py <dir-or-db> [which-minibatch=200] [marker-kernel-prefix=k_ppo_logp]
"""
import glob, os, sqlite3, sys


def main(path, which=200, marker="k_ppo_logp"):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))[-1]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    scol = "stream_id" if "stream_id" in cols else qcol
    marks = [r[0] for r in db.execute("select start from kernels where name like ? order by start", (marker + "%",))]
    if len(marks) < which + 2:
        which = len(marks) // 2
    t0, t1 = marks[which], marks[which + 1]
    rows = db.execute(f"select name, start, end, {qcol}, {scol} from kernels where start >= ? and start < ? order by start", (t0, t1)).fetchall()
    print(f"# minibatch {which}: {len(rows)} kernels, wall {(t1 - t0) / 1e3:.1f} us   (columns: start_us dur_us queue stream name)")
    busy = {}
    for name, s, e, q, st in rows:
        busy[(q, st)] = busy.get((q, st), 0) + (e - s)
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {q!s:>4} {st!s:>4}  {name[:100]}")
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
        print(f"# queue/stream {k}: busy {v / 1e3:.1f} us")
    # union of intervals
    iv = sorted((s, e) for _, s, e, _, _ in rows)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    print(f"# union of kernel intervals {tot / 1e3:.1f} us; idle (no kernel running) {(t1 - t0 - tot) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200, sys.argv[3] if len(sys.argv) > 3 else "k_ppo_logp")
