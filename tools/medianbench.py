#!/usr/bin/env python3
"""comp_bar_ohlcv without and with the median over bar lengths from 600 to 1.7e6 ticks (time bars of 30 s ... 1 day on N resident ticks).
usage: medianbench.py [N] [interval_seconds ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finmlkit_amd import _ffi, engine
ctx = _ffi.default_context()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(x) for x in sys.argv[2:]] or (30.0, 60.0, 75.0, 90.0, 120.0, 180.0, 300.0, 600.0, 1800.0, 3600.0, 14400.0, 86400.0)
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
def best(fn, reps=3):
    fn(); ctx.sync(); b = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); b = min(b, ctx.timer_stop()); del r
    return b
for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    print("interval %7.0f s %8d bars of %8d ticks: ohlcv %.2f ms, with median %.2f ms" % (iv, ci.n - 1, n // (ci.n - 1), best(lambda: t.bar_ohlcv(ci, want_median=False)), best(lambda: t.bar_ohlcv(ci, want_median=True))), flush=True)
