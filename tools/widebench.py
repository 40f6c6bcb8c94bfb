#!/usr/bin/env python3
"""Footprints / cfg 4 on a price path that covers MANY levels per bar (steps of up to `w` ticks: ~50 * w / 15 levels per 1 200-tick bar)
against the synthetic tape's own path (steps of one or two ticks: ~50 levels).  usage: widebench.py [N] [w ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
ws = [int(v) for v in sys.argv[2:]] or [0, 5, 15, 50]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
if os.environ.get("WIDE_FULL_MANTISSA"):      # full-mantissa float32 sizes: every bar takes the tick-ordered path
    import ctypes as C
    from finmlkit_amd._ffi import c_i64
    am2 = DeviceArray(ctx, n, np.float32)
    ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(42), c_i64(n), am2.p)
    t = engine.DeviceTrades(ctx, t.ts, t.price, am2, t.side)
clock, ci = t.time_bar_index(60.0)
nb = ci.n - 1
rng = np.random.default_rng(2)


def best(fn, reps=4):
    fn(); ctx.sync()
    b = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); b = min(b, ctx.timer_stop()); del r
    return b


for w in ws:
    if w == 0:
        tt, tag = t, "the tape's path"
    else:
        px = 60000.0 + 0.01 * np.cumsum(rng.integers(-w, w + 1, n)).astype(np.float64)
        px = np.round(np.maximum(px, 1.0), 2)
        tt, tag = engine.DeviceTrades(ctx, t.ts, DeviceArray.from_host(ctx, px), t.amount, t.side), f"steps of up to {w} ticks"
        del px
    o = tt.bar_ohlcv(ci, want_median=False)
    off, flat, bar, bad = tt.bar_footprints(ci, o["low"], o["high"], 0.01)
    lev = np.diff(off.to_host())
    a = best(lambda: tt.bar_footprints(ci, o["low"], o["high"], 0.01))
    b = best(lambda: tt.bars_fused(ci, 0.01, 3.0, want_median=True))
    print(f"{tag:28s}: levels per bar median {int(np.median(lev)):6d} max {int(lev.max()):7d} | footprints {a:7.2f} ms | cfg 4 {b:7.2f} ms   ({n} ticks, {nb} bars)", flush=True)
    del off, flat, bar, o
