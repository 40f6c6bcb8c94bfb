#!/usr/bin/env python3
"""Does ONE allocation for the four columns run the time-bar step at the level of the best placement?  (bench.py's headline is measured on the
columns as the library allocates them.)  Per trial: the step's dominant-kernel time (HIP events) on (a) four separate allocations, (b) one 21 GB
slab, (c) the first 21 GB of one 64 GiB slab, (d) of one 116 GiB slab; slabs are released between trials.   usage: slabcheck.py [ticks] [trials]"""
import ctypes as C, os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _ffi.default_context()


def kernel_ms(t, steps=10):
    for _ in range(4):
        t.time_bars_ohlcv(60.0)
    ctx.sync()
    ctx.call("fmk_profile_enable", C.c_int(1))
    for _ in range(steps):
        t.time_bars_ohlcv(60.0)
    k = (C.c_double * 256)(); kn = C.c_int()
    ctx.call("fmk_profile_read", k, C.c_int(256), C.byref(kn))
    ctx.call("fmk_profile_enable", C.c_int(0))
    return sum(k[i] for i in range(kn.value)) / steps


span = (21 * n + (8 << 20) + (1 << 30) - 1) // (1 << 30) * (1 << 30)
for trial in range(trials):
    row = []
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
    row.append(("separate", kernel_ms(t)))
    del t; gc.collect()
    for label, size in (("slab 21G", span), ("slab 64G", 64 << 30), ("slab 116G", 116 << 30)):
        slab = DeviceArray(ctx, size, np.uint8)
        t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx, into=(slab, 0))
        row.append((label, kernel_ms(t)))
        del t, slab; gc.collect()
    print("trial %d: " % trial + "   ".join("%s %.3f ms" % r for r in row), flush=True)
