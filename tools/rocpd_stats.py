#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 run from its rocpd SQLite output
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on this ROCm) -> CSV on stdout.
usage: rocpd_stats.py results.db [substring filter] [--last N]
--last N: only the LAST N dispatches of the kernels that match the filter (bench.py's timed region is the last steps x launches-per-step
dispatches of its dominant kernel: everything before is placement probes, settling and warm-up)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "info_kernel_symbol" in t][0]
argv = list(sys.argv)
last = 0
if "--last" in argv:
    k = argv.index("--last"); last = int(argv[k + 1]); del argv[k:k + 2]
flt = argv[2] if len(argv) > 2 else ""
if last:
    rows = list(cur.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {sym} s on d.kernel_id=s.id where s.kernel_name like ? "
                            f"order by d.start desc limit ?", (f"%{flt}%", last)))
    by = {}
    for name, dur in rows:
        by.setdefault(name, []).append(dur)
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    tot = sum(sum(v) for v in by.values()) or 1
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f'"{name}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / tot:.2f},{min(v)},{max(v)}')
    sys.exit(0)
q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
     f"from {kd} d join {sym} s on d.kernel_id=s.id where s.kernel_name like ? group by s.kernel_name order by 3 desc")
rows = list(cur.execute(q, (f"%{flt}%",)))
tot = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for name, n, t, a, mn, mx in rows:
    print(f'"{name}",{n},{t},{a:.1f},{100.0 * t / tot:.2f},{mn},{mx}')
