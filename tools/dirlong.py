#!/usr/bin/env python3
"""order-flow features on long time bars (hourly .. daily) of N resident ticks: host wall best of 3 + redo statistics.  usage: dirlong.py [N] [interval ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(x) for x in sys.argv[2:]] or [3600.0, 14400.0, 86400.0]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
if os.environ.get("FMK_FULL_MANTISSA"):                            # sizes with 24 random mantissa bits (what real sizes look like)
    import ctypes as C
    import numpy as np
    am2 = _ffi.DeviceArray(ctx, n, np.float32)
    ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(42), c_i64(n), am2.p)
    t = engine.DeviceTrades(ctx, t.ts, t.price, am2, t.side)
for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    best = 1e9
    for _ in range(4):
        ctx.sync(); t0 = time.perf_counter(); r = t.bar_directional(ci); ctx.sync(); best = min(best, (time.perf_counter() - t0) * 1e3); del r
    st = (c_i64 * 10)(); ctx.call("fmk_diag_dir_redo", st)
    print(f"n={n:.3g} {iv:g}-second bars ({ci.n - 1} bars): order flow {best:.2f} ms; redo pairs per column {list(st[3:10])}, {st[2]} of {st[1]} chunks term by term (4 calls)", flush=True)
