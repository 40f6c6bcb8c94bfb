#!/usr/bin/env python3
"""comp_bar_ohlcv (+ median) and comp_bar_trade_size_features on sizes with HEAVY TIES (decimal lots: round(lognormal, 2) + 0.01, what
real trade sizes look like) against the synthetic tape's own sizes, N ticks in time bars of the given intervals.
usage: tiesbench.py [N] [interval_s ...]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64, c_f64

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
ivs = [float(v) for v in sys.argv[2:]] or [5.0, 60.0, 600.0]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
rng = np.random.default_rng(1)
lots = (np.round(rng.lognormal(-4.5, 2.0, n), 3) + 0.001).astype(np.float32)
print("distinct sizes:", len(np.unique(lots[:1_000_000])), "of the first 1e6; share of the most common:", float(np.unique(lots[:1_000_000], return_counts=True)[1].max() / 1e6))
t2 = engine.DeviceTrades(ctx, t.ts, t.price, DeviceArray.from_host(ctx, lots), t.side)
del lots


def best(fn, reps=5):
    fn(); ctx.sync()
    b = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); b = min(b, ctx.timer_stop()); del r
    return b


for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    nb = ci.n - 1
    row = [f"interval {iv:7.0f} s {nb:9d} bars of {n // nb:7d} ticks |"]
    for name, tt in (("tape sizes", t), ("decimal lots", t2)):
        a = best(lambda: tt.bar_ohlcv(ci, want_median=False))
        b = best(lambda: tt.bar_ohlcv(ci, want_median=True))
        theta = tt.bar_ohlcv(ci, want_median=True)["median_trade_size"]
        keys = [DeviceArray(ctx, nb, np.float32) for _ in range(4)]
        c = best(lambda: ctx.call("fmk_comp_bar_trade_size_dev", tt.amount.p, C.c_int(tt.amount_is_f64), c_i64(tt.n), theta.p, ci.p,
                                  c_i64(ci.n), c_f64(5.0), *[k.p for k in keys]))
        d = best(lambda: tt.bars_fused(ci, 0.01, 3.0, want_median=True)) if os.environ.get("TIES_CFG4") else float("nan")
        row.append(f"{name}: ohlcv {a:6.2f}  + median {b:6.2f}  trade size {c:6.2f}  cfg 4 {d:6.2f} ms |")
    print(" ".join(row), flush=True)
