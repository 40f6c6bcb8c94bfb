#!/usr/bin/env python3
"""Per-dispatch rows of one kernel from a rocprofv3 rocpd database: duration, grid, workgroup, LDS -- to tell the size classes of a
kernel apart.  usage: rocpd_dispatches.py results.db <kernel substring> [max rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "info_kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
want = [c for c in ("grid_size_x", "workgroup_size_x", "lds_block_size", "group_segment_size") if c in cols]
q = f"select d.end-d.start, {', '.join('d.' + c for c in want)} from {kd} d join {sym} s on d.kernel_id=s.id where s.kernel_name like ? order by d.start"
rows = list(cur.execute(q, (f"%{sys.argv[2]}%",)))
print("duration_us", *want)
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{r[0] / 1e3:10.1f}", *r[1:])
