#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 run from its rocpd SQLite output
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on this ROCm) -> CSV on stdout.
usage: rocpd_stats.py results.db [substring filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "info_kernel_symbol" in t][0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
     f"from {kd} d join {sym} s on d.kernel_id=s.id where s.kernel_name like ? group by s.kernel_name order by 3 desc")
rows = list(cur.execute(q, (f"%{flt}%",)))
tot = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for name, n, t, a, mn, mx in rows:
    print(f'"{name}",{n},{t},{a:.1f},{100.0 * t / tot:.2f},{mn},{mx}')
