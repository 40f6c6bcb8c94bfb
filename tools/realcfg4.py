#!/usr/bin/env python3
"""cfg 4 (bars_fused) on lognormal bar lengths with full-mantissa sizes only -- for kernel profiles.  usage: realcfg4.py [N] [sigma] [dyadic]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
sg = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dy = len(sys.argv) > 3
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
if not dy:
    am2 = DeviceArray(ctx, n, np.float32)
    ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(42), c_i64(n), am2.p)
    t = engine.DeviceTrades(ctx, t.ts, t.price, am2, t.side)
rng = np.random.default_rng(7)
nb0 = n // 1200
lens = np.maximum(1, rng.lognormal(np.log(1200.0) - sg * sg / 2, sg, int(nb0 * 1.3)).astype(np.int64)) if sg > 0 else np.full(nb0, 1200, np.int64)
ci_h = np.concatenate([[-1], np.cumsum(lens) - 1]); ci_h = ci_h[ci_h <= n - 1].astype(np.int64)
ci = DeviceArray.from_host(ctx, ci_h)
import os
if os.environ.get("ONLY_FP_IV"):
    clock, ci = t.time_bar_index(float(os.environ["ONLY_FP_IV"]))
if os.environ.get("ONLY_FP") or os.environ.get("ONLY_FP_IV"):
    o = t.bar_ohlcv(ci, want_median=True)
    for _ in range(2):
        ctx.sync(); s0 = time.perf_counter(); r = t.bar_footprints(ci, o["low"], o["high"], 0.01); ctx.sync(); print(f"footprints: {(time.perf_counter() - s0) * 1e3:.2f} ms"); del r
    sys.exit(0)
all_ms = []
for _ in range(6):
    ctx.sync(); s = time.perf_counter(); r = t.bars_fused(ci, 0.01, 3.0); ctx.sync(); all_ms.append(round((time.perf_counter() - s) * 1e3, 2)); del r
print(f"cfg 4, sigma {sg}, {'dyadic' if dy else 'full-mantissa'}: {all_ms} ms")
o = t.bar_ohlcv(ci, want_median=True)
for _ in range(3):
    ctx.sync(); s = time.perf_counter(); r = t.bar_footprints(ci, o["low"], o["high"], 0.01); ctx.sync(); ms = (time.perf_counter() - s) * 1e3; del r
print(f"footprints alone: {ms:.2f} ms")
for _ in range(3):
    ctx.sync(); s = time.perf_counter(); r = t.bar_directional(ci); ctx.sync(); ms = (time.perf_counter() - s) * 1e3; del r
print(f"order flow alone: {ms:.2f} ms")
for _ in range(3):
    ctx.sync(); s = time.perf_counter(); r = t.bar_ohlcv(ci, want_median=True); ctx.sync(); ms = (time.perf_counter() - s) * 1e3; del r
print(f"ohlcv + median alone: {ms:.2f} ms")
