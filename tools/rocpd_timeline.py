#!/usr/bin/env python3
"""The last N kernel dispatches (and memory copies, when traced) of a rocprofv3 rocpd database in time order: start offset, duration,
gap to the previous one -- to see what a step spends between its kernels.  usage: rocpd_timeline.py results.db [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {sym} s on d.kernel_id=s.id order by d.start"))
mc = [t for t in tabs if "memory_copy" in t and "info" not in t]
if mc:
    try:
        rows += [(a, b, "memcpy") for a, b in cur.execute(f"select start, end from {mc[0]}")]
    except sqlite3.Error:
        pass
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = rows[0][0]
prev = None
for a, b, name in rows:
    gap = (a - prev) / 1e3 if prev else 0.0
    print(f"{(a - t0) / 1e3:10.1f} us  dur {(b - a) / 1e3:9.1f} us  gap {gap:7.1f} us  {name[:70]}")
    prev = b
