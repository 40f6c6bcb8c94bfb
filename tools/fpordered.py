#!/usr/bin/env python3
"""comp_bar_footprints (size + fill) and trade-size / order-flow features on amounts with a FULL random float32 mantissa (what real
decimal lots look like after TradesData's float32 cast): every float32 level sum rounds, so every bar takes the tick-ordered
accumulation.  usage: fpordered.py [N] [interval_seconds ...]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(x) for x in sys.argv[2:]] or (1.0, 10.0, 60.0, 600.0, 3600.0, 86400.0)
ctx = _ffi.default_context()
t0 = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
am2 = DeviceArray(ctx, n, np.float32)
ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(42), c_i64(n), am2.p)
t = engine.DeviceTrades(ctx, t0.ts, t0.price, am2, t0.side)


def best(fn, reps=2):
    b = 1e9
    for _ in range(reps):
        ctx.sync(); s = time.perf_counter(); r = fn(); ctx.sync(); b = min(b, (time.perf_counter() - s) * 1e3); del r
    return b


for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    nb = ci.n - 1
    o = t.bar_ohlcv(ci, want_median=True)
    ms_f = best(lambda: t.bar_footprints(ci, o["low"], o["high"], 0.01))
    ms_d = best(lambda: t.bar_directional(ci))
    ms_c = best(lambda: t.bars_fused(ci, 0.01, 3.0))
    print(f"full-mantissa float32 amounts, interval {iv:8.0f} s: {nb:9d} bars of {n // max(nb, 1):8d} ticks | footprints {ms_f:7.2f} | "
          f"order flow {ms_d:7.2f} | cfg 4 {ms_c:7.2f} ms", flush=True)
    del o, clock, ci
