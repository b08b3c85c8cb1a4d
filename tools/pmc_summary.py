"""Per-kernel average of rocprofv3 --pmc counters from a rocpd sqlite db (prints a small table)."""
import glob, os, sqlite3, sys


def main(path, pattern="k_"):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))[-1]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    # counters_collection view: one row per (dispatch, counter)
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    q = f"select {name_col}, counter_name, avg(value), count(*) from counters_collection group by {name_col}, counter_name"
    rows = db.execute(q).fetchall()
    out = {}
    for k, c, v, n in rows:
        if pattern in k:
            out.setdefault((k[5:] if k.startswith("void ") else k).split("(")[0].replace("anonymous namespace)::", "")[:60] or k[:60], {})[c] = (v, n)
    for k, d in out.items():
        print(k)
        for c, (v, n) in sorted(d.items()):
            print(f"    {c:28s} {v:16.1f}   (n={n})")


if __name__ == "__main__":
    main(*sys.argv[1:])
