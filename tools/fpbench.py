#!/usr/bin/env python3
"""Kernel-only timing of comp_bar_footprints (fill phase) with preallocated outputs."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, FootprintOut, FOOTPRINT_FLAT_FIELDS, FOOTPRINT_BAR_FIELDS, c_i64, c_f64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
if len(sys.argv) > 2 and sys.argv[2] == "inexact":       # non-dyadic amounts: forces the tick-ordered path
    rng = np.random.default_rng(0)
    blk = rng.lognormal(-1, 1.0, 1 << 22).astype(np.float32)
    host = np.tile(blk, n // len(blk) + 1)[:n]
    t.amount = DeviceArray.from_host(ctx, host)
clock, ci = t.time_bar_index(60.0)
o = t.bar_ohlcv(ci, False)
nb = ci.n - 1
off = DeviceArray(ctx, nb + 1, np.int64)
tot, mx = c_i64(), c_i64()
ctx.call("fmk_comp_bar_footprints_size_dev", o["low"].p, o["high"].p, c_i64(nb), c_f64(0.01), off.p, C.byref(tot), C.byref(mx))
flat = {k: DeviceArray(ctx, tot.value, dt) for k, dt in FOOTPRINT_FLAT_FIELDS}
bar = {k: DeviceArray(ctx, nb, dt) for k, dt in FOOTPRINT_BAR_FIELDS}
st = FootprintOut(**{k: v.ptr for k, v in {**flat, **bar}.items()})
bad = DeviceArray(ctx, 1, np.int64); bad.zero()
def run():
    ctx.call("fmk_comp_bar_footprints_fill_dev", t.price.p, t.amount.p, C.c_int(t.amount_is_f64), c_i64(n), ci.p,
             c_i64(ci.n), t.side.p, c_f64(0.01), o["low"].p, c_f64(3.0), off.p, c_i64(mx.value), C.byref(st), bad.p)
run(); ctx.sync()
ts = []
for _ in range(5):
    ctx.timer_start(); run(); ts.append(ctx.timer_stop())
print(f"footprints fill n={n} levels={tot.value} max_levels={mx.value} ordered={os.environ.get('FMK_FP_ORDERED','0')} "
      f"mode={'inexact' if len(sys.argv)>2 else 'exact'}: {np.median(ts):.3f} ms")
