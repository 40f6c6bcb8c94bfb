#!/usr/bin/env python3
"""Does the level of the dominant kernel on an UNTOUCHED set of columns move when other memory is allocated or freed?  One set of columns
(synthesised once), the kernel timed by the library's HIP events after each of: allocate + zero 64 GiB (B), allocate + zero 64 GiB more (C),
free B, free C -- two cycles.  usage: placeshift.py [ticks]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=1, first=0, ctx=ctx)
clock, idx = t.time_bar_index(60.0)
out = t.alloc_ohlcv(idx.n - 1, True)
GiB = 1 << 30

def level(reps=12, warm=6):
    for _ in range(warm):
        t.bar_ohlcv(idx, want_median=True, out=out)
    ctx.sync()
    ctx.call("fmk_profile_enable", C.c_int(1))
    for _ in range(reps):
        t.bar_ohlcv(idx, want_median=True, out=out)
    ctx.sync()
    kms = (C.c_double * 256)(); kn = C.c_int()
    ctx.call("fmk_profile_read", kms, C.c_int(256), C.byref(kn))
    ctx.call("fmk_profile_enable", C.c_int(0))
    v = sorted(kms[i] for i in range(kn.value))
    return v[len(v) // 2]

print(f"columns: price {t.price.ptr:#x} amount {t.amount.ptr:#x}")
print(f"start                      : {level():.3f} ms", flush=True)
for cyc in range(2):
    B = DeviceArray(ctx, 64 * GiB, np.uint8); B.zero(); ctx.sync()
    print(f"cycle {cyc}: + 64 GiB (B {B.ptr:#x}) : {level():.3f} ms", flush=True)
    Cc = DeviceArray(ctx, 64 * GiB, np.uint8); Cc.zero(); ctx.sync()
    print(f"cycle {cyc}: + 64 GiB (C {Cc.ptr:#x}) : {level():.3f} ms", flush=True)
    B.free(); ctx.trim(); ctx.sync()
    print(f"cycle {cyc}: B freed                : {level():.3f} ms", flush=True)
    Cc.free(); ctx.trim(); ctx.sync()
    print(f"cycle {cyc}: C freed                : {level():.3f} ms", flush=True)
print(f"again                      : {level():.3f} ms")
