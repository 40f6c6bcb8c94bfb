#!/usr/bin/env python3
"""Timing of the tick-level volatility loops at N ticks (HIP events, outputs preallocated by the library calls).
   python tools/tlbench.py [N] [ew]     ew: only the exponentially weighted ones"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
ctx.sync()


def timed(name, fn, reps=3):
    fn(); ctx.sync()
    best = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); ms = ctx.timer_stop(); best = min(best, ms); del r
    print(f"{name:32s} {best:8.2f} ms", flush=True)


r = t.lagged_returns(5.0, True)
only_ew = len(sys.argv) > 2 and sys.argv[2] == "ew"
if not only_ew:
    timed("lagged_returns 5s log", lambda: t.lagged_returns(5.0, True))
timed("ewmst 60s", lambda: t.ewmst(r, 60.0))
timed("ewmst_mean0 60s", lambda: t.ewmst(r, 60.0, mean0=True))
timed("ewms span 100", lambda: t.ewms(r, 100))
for w in () if only_ew else (20, 64, 100, 256, 500, 1000, 2048, 4096, 100_000):
    timed(f"realized_vol window {w}", lambda: t.realized_vol(r, w, True))
