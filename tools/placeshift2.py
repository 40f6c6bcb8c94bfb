#!/usr/bin/env python3
"""Is the placement effect an interplay of the two streams the dominant kernel reads (price, amount)?  At K positions of one slab (16 GiB apart) the columns
are laid out as usual, except that the AMOUNT column is moved by delta bytes (0, 64 KiB ... 1 GiB); the step's dominant-kernel time (HIP events of the
library) for every (position, delta).  A delta that flattens the positions would be a layout the library could simply choose.
usage: placeshift2.py [ticks] [positions]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = _ffi.default_context()
step = 16 << 30
deltas = [0, 64 << 10, 192 << 10, 1 << 20, (2 << 20) + (64 << 10), 32 << 20, 1 << 30]
span = 21 * n + (16 << 20) + (2 << 30)
slab = DeviceArray(ctx, (K - 1) * step + span, np.uint8)
al = lambda x: (x + (2 << 20) - 1) // (2 << 20) * (2 << 20)


def kernel_ms(t, steps=6):
    for _ in range(3):
        t.time_bars_ohlcv(60.0)
    ctx.sync()
    ctx.call("fmk_profile_enable", C.c_int(1))
    for _ in range(steps):
        t.time_bars_ohlcv(60.0)
    k = (C.c_double * 256)(); kn = C.c_int()
    ctx.call("fmk_profile_read", k, C.c_int(256), C.byref(kn))
    ctx.call("fmk_profile_enable", C.c_int(0))
    return sum(k[i] for i in range(kn.value)) / steps


print("delta:      " + "  ".join(f"{d >> 10:>8d}K" for d in deltas))
for i in range(K):
    base = i * step
    row = []
    for d in deltas:
        o_ts = base
        o_px = o_ts + al(8 * n)
        o_am = o_px + al(8 * n) + d
        o_sd = o_am + al(4 * n) + (2 << 20)
        cols = [DeviceArray(ctx, n, dt, slab.ptr + o, owner=slab) for dt, o in ((np.int64, o_ts), (np.float64, o_px), (np.float32, o_am), (np.int8, o_sd))]
        ctx.call("fmk_synth_trades_dev", C.c_uint64(42), c_i64(0), c_i64(n), C.c_uint64(engine.DENSE_GAP_MOD), *[c.p for c in cols])
        t = engine.DeviceTrades(ctx, *cols)
        row.append(kernel_ms(t))
        del t, cols
    print(f"position {i}: " + "  ".join(f"{m:9.3f}" for m in row), flush=True)
