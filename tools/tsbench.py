#!/usr/bin/env python3
"""comp_bar_trade_size_features on N resident ticks in time bars of the given intervals.  usage: tsbench.py [N] [interval_s ...]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64, c_f64

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(v) for v in sys.argv[2:]] or [60.0]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    nb = ci.n - 1
    theta = t.bar_ohlcv(ci, want_median=True)["median_trade_size"]
    keys = ("mean_size_rel", "size_95_rel", "pct_block", "size_gini")
    out = {k: DeviceArray(ctx, nb, np.float32) for k in keys}
    fn = lambda: ctx.call("fmk_comp_bar_trade_size_dev", t.amount.p, C.c_int(t.amount_is_f64), c_i64(t.n), theta.p, ci.p,
                          c_i64(ci.n), c_f64(5.0), *[out[k].p for k in keys])
    fn(); ctx.sync()
    b = 1e9
    for _ in range(5):
        ctx.timer_start(); fn(); b = min(b, ctx.timer_stop())
    chk = [float(np.nansum(out[k].to_host().astype(np.float64))) for k in keys]
    print(f"interval {iv:8.0f} s  {nb:9d} bars of {n // nb:8d} ticks: {b:8.2f} ms   checksums {chk}", flush=True)
