"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a text table.

    python tools/rocprof_summary.py gpurun_out/prof_x/<host>/<pid>_results.db [> profiles/rNN_x.txt]
"""
import glob
import os
import sqlite3
import sys


def summarise(path, out=sys.stdout, top=40):
    if os.path.isdir(path):
        c = sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))
        if not c:
            raise SystemExit("no *_results.db under " + path)
        path = c[-1]
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                      "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 kernel-trace summary: {os.path.basename(path)}", file=out)
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
          f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scr':>5s} {'grid_x':>9s} {'wg':>5s}", file=out)
    for r in rows[:top]:
        name = r[0] if len(r[0]) <= 70 else r[0][:67] + "..."
        print(f"{name:70s} {r[1]:7d} {r[2]/1e6:10.3f} {r[3]/1e3:10.2f} {r[4]/1e3:9.2f} {r[5]/1e3:9.2f} {100*r[2]/tot:6.2f} "
              f"{r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:5d} {r[11]:9d} {r[12]:5d}", file=out)
    print(f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches", file=out)


if __name__ == "__main__":
    summarise(sys.argv[1])
