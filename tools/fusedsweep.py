#!/usr/bin/env python3
"""cfg 4's call (bars_fused: OHLCV + median, order flow, footprints) over time bars of 1 s ... 1 day on N resident ticks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finmlkit_amd import _ffi, engine
ctx = _ffi.default_context()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
def best(fn, reps=2):
    fn(); ctx.sync(); b = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); b = min(b, ctx.timer_stop()); del r
    return b
for iv in (1.0, 10.0, 60.0, 120.0, 600.0, 3600.0, 86400.0):
    clock, ci = t.time_bar_index(iv)
    print("interval %7.0f s %9d bars of %8d ticks: bars_fused %.2f ms" % (iv, ci.n - 1, n // (ci.n - 1), best(lambda: t.bars_fused(ci, 0.01, 3.0))), flush=True)
