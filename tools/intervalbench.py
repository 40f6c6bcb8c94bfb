#!/usr/bin/env python3
"""The bar reducers over bar lengths from 20 ticks to 1.7e6 ticks (time bars of 1 s ... 1 day on N resident ticks): OHLCV + median,
order flow, trade-size features, footprints (size + fill).  One line per interval.  usage: intervalbench.py [N] [interval_seconds ...]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64, c_f64

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(x) for x in sys.argv[2:]] or (1.0, 10.0, 60.0, 600.0, 3600.0, 86400.0)
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)


def best(fn, reps=3):
    fn(); ctx.sync()
    b = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); b = min(b, ctx.timer_stop()); del r
    return b


for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    nb = ci.n - 1
    o = t.bar_ohlcv(ci, want_median=True)
    ms_o = best(lambda: t.bar_ohlcv(ci, want_median=True))
    ms_d = best(lambda: t.bar_directional(ci))
    st = (c_i64 * 10)()
    ctx.call("fmk_diag_dir_redo", st)
    ctx.sync(); t.bar_directional(ci); ctx.call("fmk_diag_dir_redo", st)
    keys = [DeviceArray(ctx, nb, np.float32) for _ in range(4)]
    ms_t = best(lambda: ctx.call("fmk_comp_bar_trade_size_dev", t.amount.p, C.c_int(t.amount_is_f64), c_i64(n), o["median_trade_size"].p,
                                 ci.p, c_i64(ci.n), c_f64(5.0), *[k.p for k in keys]))
    ms_f = best(lambda: t.bar_footprints(ci, o["low"], o["high"], 0.01), reps=2)
    ms_c = best(lambda: t.bars_fused(ci, 0.01, 3.0), reps=2)
    print(f"interval {iv:8.0f} s: {nb:9d} bars of {n // max(nb, 1):8d} ticks | ohlcv+median {ms_o:6.2f} | order flow {ms_d:6.2f} | "
          f"trade size {ms_t:6.2f} | footprints (size + fill + allocation) {ms_f:6.2f} | cfg 4 (bars_fused) {ms_c:6.2f} ms", flush=True)
    print(f"      order-flow redo: {st[0]} (bar, column) pairs, {st[2]} of {st[1]} chunks term by term; per column {list(st[3:10])}", flush=True)
    del o, keys, clock, ci
