#!/usr/bin/env python3
"""Does taking the four input columns from ONE allocation change the step-time lottery (tools/allocwarm.py: the level is
a property of the input columns' allocation)?  One process, alternating rounds: (a) four separate allocations, as
engine.DeviceTrades.synth makes them; (b) one arena, columns at 2 MiB-aligned offsets.  Each round frees everything, gives
the pooled blocks back to the driver, allocates, fills, and times 20 steps of time-bar OHLCV + median.
usage: arenawarm.py [N] [rounds]"""
import ctypes as C, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = _ffi.default_context()
ALIGN = 2 << 20


def arena_trades():
    sizes = [8 * n, 8 * n, 4 * n, n]
    offs, tot = [], 0
    for s in sizes:
        offs.append(tot)
        tot += (s + ALIGN - 1) // ALIGN * ALIGN
    arena = DeviceArray(ctx, tot, np.uint8)
    cols = [DeviceArray(ctx, n, dt, arena.ptr + o, owner=arena) for dt, o in zip((np.int64, np.float64, np.float32, np.int8), offs)]
    ctx.call("fmk_synth_trades_dev", C.c_uint64(42), c_i64(0), c_i64(n), C.c_uint64(engine.DENSE_GAP_MOD),
             cols[0].p, cols[1].p, cols[2].p, cols[3].p)
    t = engine.DeviceTrades(ctx, *cols)
    t._backing = [arena]
    return t


def measure(t):
    clock, ci = t.time_bar_index(60.0)
    o = t.alloc_ohlcv(ci.n - 1, True)
    for _ in range(3):
        t.bar_ohlcv(ci, True, out=o)
    ctx.sync()
    ms = []
    for _ in range(20):
        ctx.timer_start(); t.bar_ohlcv(ci, True, out=o); ms.append(ctx.timer_stop())
    return float(np.mean(ms))


res = {"separate": [], "arena": []}
for r in range(rounds):
    for kind in ("separate", "arena"):
        gc.collect(); ctx.trim()
        t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx) if kind == "separate" else arena_trades()
        res[kind].append(measure(t))
        del t
for k, v in res.items():
    print("%-9s %s   mean %.3f  min %.3f  max %.3f" % (k, " ".join("%.3f" % x for x in v), np.mean(v), min(v), max(v)))
