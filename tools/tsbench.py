#!/usr/bin/env python3
"""comp_bar_trade_size_features over time bars of several lengths on N ticks (theta = each bar's median trade size, what the kits pass):
host wall, best of 3.  usage: tsbench.py [N] [interval_seconds ...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(x) for x in sys.argv[2:]] or [1.0, 3.0, 10.0, 30.0, 60.0, 120.0, 600.0]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    o = t.bar_ohlcv(ci)
    nb = ci.n - 1
    outs = [DeviceArray(ctx, nb, np.float32) for _ in range(4)]
    best = 1e9
    for _ in range(3):
        ctx.sync(); t0 = time.perf_counter()
        ctx.call("fmk_comp_bar_trade_size_dev", t.amount.p, C.c_int(t.amount_is_f64), c_i64(n), o["median_trade_size"].p, ci.p,
                 c_i64(ci.n), C.c_double(5.0), *[x.p for x in outs])
        ctx.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"n={n:.3g} {iv:g}-second bars ({n / nb:.0f} ticks/bar): trade-size features {best:.2f} ms "
          f"(FMK_TS_LANES={os.environ.get('FMK_TS_LANES', 'default')})", flush=True)
    del o, outs, clock, ci
