#!/usr/bin/env python3
"""cfg 3 alone for a kernel trace: the volume- or dollar-bar indexer (default exact mode) on synthetic ticks, `reps` calls.
usage: dollarprof.py [ticks] [reps] [dollar|volume]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
which = sys.argv[3] if len(sys.argv) > 3 else "dollar"
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
clock, ci = t.time_bar_index(60.0)
o = t.bar_ohlcv(ci, want_median=False)
vol_total = float(o["volume"].to_host().astype(np.float64).sum())
span_days = (t.first_last_ts()[1] - t.first_last_ts()[0]) / 86400e9
vthr = vol_total / max(span_days, 1e-9) / 2000.0
dthr = vthr * float(np.median(o["close"].to_host()))
del o
fn = (lambda: t.dollar_bar_index(dthr)) if which == "dollar" else (lambda: t.volume_bar_index(vthr))
fn(); ctx.sync()
print("MARK timed calls start", flush=True)
import time
t0 = time.perf_counter()
for _ in range(reps):
    fn()
ctx.sync()
print(f"{which}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per call, uncertified {t.last_uncertified}")
