#!/usr/bin/env python3
"""Kernel-only timings of the cfg-4 reducers at N ticks: OHLCV, directional, footprint fill, fused fill.
usage: flowbench.py [N] [inexact]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import (DeviceArray, DirectionalOut, FootprintOut, DIRECTIONAL_FIELDS, FOOTPRINT_FLAT_FIELDS,
                               FOOTPRINT_BAR_FIELDS, c_i64, c_f64)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
if len(sys.argv) > 2 and sys.argv[2] == "inexact":       # non-dyadic amounts: forces the tick-ordered path
    rng = np.random.default_rng(0)
    blk = rng.lognormal(-1, 1.0, 1 << 22).astype(np.float32)
    t.amount = DeviceArray.from_host(ctx, np.tile(blk, n // len(blk) + 1)[:n])
clock, ci = t.time_bar_index(60.0)
o = t.bar_ohlcv(ci, True)
nb = ci.n - 1
off = DeviceArray(ctx, nb + 1, np.int64)
tot, mx = c_i64(), c_i64()
ctx.call("fmk_comp_bar_footprints_size_dev", o["low"].p, o["high"].p, c_i64(nb), c_f64(0.01), off.p, C.byref(tot), C.byref(mx))
flat = {k: DeviceArray(ctx, tot.value, dt) for k, dt in FOOTPRINT_FLAT_FIELDS}
bar = {k: DeviceArray(ctx, nb, dt) for k, dt in FOOTPRINT_BAR_FIELDS}
fst = FootprintOut(**{k: v.ptr for k, v in {**flat, **bar}.items()})
d = {k: DeviceArray(ctx, nb, dt) for k, dt in DIRECTIONAL_FIELDS}
dst = DirectionalOut(**{k: d[k].ptr for k in d})
cn = DeviceArray(ctx, 2, np.int64); cn.zero()
A = (t.price.p, t.amount.p, C.c_int(t.amount_is_f64), c_i64(n), ci.p, c_i64(ci.n))
runs = {
    "ohlcv+median": lambda: t.bar_ohlcv(ci, True, out=o),
    "directional": lambda: ctx.call("fmk_comp_bar_directional_dev", *A, t.side.p, C.byref(dst), cn.view(0, 1).p),
    "footprint fill": lambda: ctx.call("fmk_comp_bar_footprints_fill_dev", *A, t.side.p, c_f64(0.01), o["low"].p, c_f64(3.0),
                                       off.p, c_i64(mx.value), C.byref(fst), cn.view(1, 1).p),
}
print(f"n={n} bars={nb} levels={tot.value} max_levels={mx.value} "
      f"amounts={'inexact' if len(sys.argv) > 2 else 'dyadic'}")
for name, fn in runs.items():
    fn(); ctx.sync()
    ts = []
    for _ in range(5):
        ctx.timer_start(); fn(); ts.append(ctx.timer_stop())
    print(f"  {name:22s} {np.median(ts):8.3f} ms", flush=True)
