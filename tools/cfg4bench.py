#!/usr/bin/env python3
"""cfg 4 (bars_fused: OHLCV + order-flow + footprints) at N ticks, dyadic and full-mantissa amounts, host wall time best of 5.
usage: cfg4bench.py [N] [bar interval in seconds, default 60]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
interval = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
clock, ci = t.time_bar_index(interval)
am2 = DeviceArray(ctx, n, np.float32)
ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(42), c_i64(n), am2.p)
t2 = engine.DeviceTrades(ctx, t.ts, t.price, am2, t.side)
for name, tr in (("dyadic amounts", t), ("full-mantissa amounts", t2)):
    best = 1e9
    for _ in range(5):
        ctx.sync(); t0 = time.perf_counter(); r = tr.bars_fused(ci, 0.01, 3.0); ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3); del r
    fb = c_i64()
    ctx.call("fmk_diag_fp_median_fallbacks", C.byref(fb))
    nfp, ndir, nredo = c_i64(), c_i64(), c_i64()
    ctx.call("fmk_diag_fused_last", C.byref(nfp), C.byref(ndir), C.byref(nredo))
    z10 = (c_i64 * 10)()
    ctx.call("fmk_diag_dir_redo", z10)                  # (reads and clears)
    ctx.timer_start(); r = tr.bars_fused(ci, 0.01, 3.0); dev_ms = ctx.timer_stop(); del r
    ctx.call("fmk_diag_dir_redo", z10)
    print("  tick-order redo of that call: pairs, tiles, term-by-term tiles, pairs of column 0..6 =", list(z10), flush=True)
    print(f"  device time {dev_ms:.3f} ms; one-pass kernel (FMK_FUSED={os.environ.get('FMK_FUSED', 'unset')}): {nfp.value} bars to the footprint classes, {ndir.value} to k_bar_dir, {nredo.value} redo entries", flush=True)
    print(f"n={n:.3g} interval {interval:g} s ({n // max(ci.n - 1, 1)} ticks per bar) cfg4 {name}: {best:.3f} ms ("
          f"median deferred to the footprint sweep={os.environ.get('FMK_FLOW_MEDIAN_DEFER', '0')}, bracket misses {fb.value} of {ci.n - 1} bars)", flush=True)
