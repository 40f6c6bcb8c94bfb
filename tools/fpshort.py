#!/usr/bin/env python3
"""build_ohlcv / build_directional_features / build_footprints kernels at N ticks for several bar intervals (short bars are the
stress case of the wave-per-bar schedules).  usage: fpshort.py [N] [interval_s ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(x) for x in sys.argv[2:]] or [1.0, 10.0, 60.0]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); r = fn(); ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3); del r
    return best


for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    o = t.bar_ohlcv(ci, want_median=False)
    ms_o = timed(lambda: t.bar_ohlcv(ci))
    ms_d = timed(lambda: t.bar_directional(ci))
    ms_f = timed(lambda: t.bar_footprints(ci, o["low"], o["high"], 0.01, 3.0))
    ms_all = timed(lambda: t.bars_fused(ci, 0.01, 3.0))
    print(f"n={n:.3g} interval {iv:g}s ({ci.n - 1} bars): ohlcv+median {ms_o:.2f} ms, directional {ms_d:.2f} ms, "
          f"footprints {ms_f:.2f} ms, bars_fused {ms_all:.2f} ms", flush=True)
    del o, clock, ci
