#!/usr/bin/env python3
"""Per-step time of the bench's hot path (time-bar index + OHLCV + median at N ticks) for the FIRST process on a box:
does the first handful of steps run slower than the steady state?   usage: stepwarm.py [N] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ctx = _ffi.default_context()
t0 = time.time()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
ctx.sync()
print("synth %.3f s" % (time.time() - t0))
clock, ci = t.time_bar_index(60.0)
o = t.alloc_ohlcv(ci.n - 1, True)
ms = []
for _ in range(steps):
    ctx.timer_start(); t.bar_ohlcv(ci, True, out=o); ms.append(ctx.timer_stop())
ms = np.array(ms)
print("per-step ms:", " ".join("%.3f" % x for x in ms))
for a, b in ((0, 2), (2, 12), (12, 30), (30, steps)):
    print("steps %2d..%2d: mean %.3f  min %.3f  max %.3f" % (a, b - 1, ms[a:b].mean(), ms[a:b].min(), ms[a:b].max()))
