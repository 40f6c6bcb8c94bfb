#!/usr/bin/env python3
"""The bar reducers on a stream whose bars have HEAVY-TAILED lengths (lognormal, mean ~1 200 ticks: quiet minutes of 100 ticks next
to busy ones of 20 000, as real one-minute bars are), against the uniform 1 200-tick bars of the synthetic tape's own clock.
usage: realbars.py [N] [sigma ...]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64, c_f64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
sigmas = [float(x) for x in sys.argv[2:]] or [0.0, 0.5, 1.0, 1.5]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
am2 = DeviceArray(ctx, n, np.float32)
ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(42), c_i64(n), am2.p)
t2 = engine.DeviceTrades(ctx, t.ts, t.price, am2, t.side)


def best(fn, reps=3):
    b = 1e9
    for _ in range(reps):
        ctx.sync(); s = time.perf_counter(); r = fn(); ctx.sync(); b = min(b, (time.perf_counter() - s) * 1e3); del r
    return b


rng = np.random.default_rng(7)
for sg in sigmas:
    nb0 = n // 1200
    lens = np.maximum(1, rng.lognormal(np.log(1200.0) - sg * sg / 2, sg, int(nb0 * 1.3)).astype(np.int64)) if sg > 0 else np.full(nb0, 1200, np.int64)
    ci_h = np.concatenate([[-1], np.cumsum(lens) - 1])
    ci_h = ci_h[ci_h <= n - 1].astype(np.int64)
    ci = DeviceArray.from_host(ctx, ci_h)
    nb = ci.n - 1
    ln = np.diff(ci_h)
    o = t.bar_ohlcv(ci, want_median=True)
    ms_o = best(lambda: t.bar_ohlcv(ci, want_median=True))
    ms_d = best(lambda: t.bar_directional(ci))
    keys = [DeviceArray(ctx, nb, np.float32) for _ in range(4)]
    ms_t = best(lambda: ctx.call("fmk_comp_bar_trade_size_dev", t.amount.p, C.c_int(t.amount_is_f64), c_i64(n), o["median_trade_size"].p,
                                 ci.p, c_i64(ci.n), c_f64(5.0), *[k.p for k in keys]))
    ms_f = best(lambda: t.bar_footprints(ci, o["low"], o["high"], 0.01), reps=2)
    ms_f2 = best(lambda: t2.bar_footprints(ci, o["low"], o["high"], 0.01), reps=2)
    ms_c = best(lambda: t.bars_fused(ci, 0.01, 3.0), reps=2)
    ms_c2 = best(lambda: t2.bars_fused(ci, 0.01, 3.0), reps=2)
    print(f"lognormal bar lengths, sigma {sg:3.1f}: {nb:8d} bars, ticks/bar median {int(np.median(ln))} p99 {int(np.percentile(ln, 99))} max {ln.max()} | "
          f"ohlcv+median {ms_o:6.2f} | order flow {ms_d:6.2f} | trade size {ms_t:6.2f} | footprints {ms_f:6.2f} (full-mantissa sizes {ms_f2:6.2f}) | cfg 4 {ms_c:6.2f} "
          f"(full-mantissa sizes {ms_c2:6.2f}) ms", flush=True)
    del o, keys, ci
