#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (this rocprofv3 writes <name>_results.db unless --output-format csv is given).
usage: python tools/rocpd_stats.py gpurun_out/convprof/conv_results.db [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*) c, avg(d.end-d.start)/1e3 a, min(d.end-d.start)/1e3 from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name order by c*a desc")
rows = db.execute(q).fetchall()
total = sum(c * a for _, c, a, _ in rows)
for n, c, a, m in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{n[:72]:72s} calls {c:5d}  avg {a:8.1f} us  min {m:8.1f} us  {100 * c * a / total:5.1f} %")
