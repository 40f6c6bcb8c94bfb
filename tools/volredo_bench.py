#!/usr/bin/env python3
"""Cost of the exact-`volume` redo on the float64-amount OHLCV path: comp_bar_ohlcv without median on
NB bars of 2048 ticks, (a) amounts constructed so that EVERY bar total is a float32 tie (every bar on the redo list:
worst case), (b) random lognormal float64 amounts of the same shape (no bar expected on the list: tie test + empty redo
launch only), (c) the float32 cast of (b) (kernels without the tie test).  Run under rocprofv3 --kernel-trace --stats
for the per-kernel split.   usage: volredo_bench.py [NB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray

nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
L = 2048
n = nb * L + 1
rng = np.random.default_rng(7)
m = rng.integers(1, 3_000_000, size=(nb, L))
tot = m[:, :-1].sum(axis=1)
m[:, -1] = ((tot + 1_500_000) // 250) * 250 + 125 - tot          # bar total: odd multiple of 0.125 in [2^21, 2^22)
ties = np.concatenate([[1.0], m.reshape(-1).astype(np.float64) * 0.001])
del m
rnd = rng.lognormal(7.0, 1.0, n)
px = 100.0 + 0.01 * rng.integers(0, 500, size=n)
ts = np.arange(n, dtype=np.int64)
ctx = _ffi.default_context()
ci = DeviceArray.from_host(ctx, L * np.arange(nb + 1, dtype=np.int64))
print(f"bars={nb} ticks/bar={L} n={n}")
for name, am in (("f64 all-ties", ties), ("f64 lognormal", rnd), ("f32 lognormal", rnd.astype(np.float32))):
    t = engine.DeviceTrades.from_numpy(ts, px, am, ctx=ctx)
    o = t.bar_ohlcv(ci, False)
    ctx.sync()
    tm = []
    for _ in range(7):
        ctx.timer_start(); t.bar_ohlcv(ci, False, out=o); tm.append(ctx.timer_stop())
    vol = o["volume"].to_host()
    print(f"  {name:14s} {np.median(tm):8.3f} ms  (min {min(tm):.3f} max {max(tm):.3f})  checksum {float(vol.astype(np.float64).sum()):.6e}", flush=True)
    del t, o
