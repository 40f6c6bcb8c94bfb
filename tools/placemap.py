#!/usr/bin/env python3
"""Read-bandwidth map of one large slab (2 GiB windows, 8 B loads per lane), twice, then the dominant kernel with the price / amount
columns placed at several positions of the same slab: does a slow kernel level coincide with a slow region of the map?
usage: placemap.py [slab GiB] [ticks]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**9
ctx = _ffi.default_context()
GiB = 1 << 30
ref = engine.DeviceTrades.synth(n, seed=1, first=0, ctx=ctx)
clock, idx = ref.time_bar_index(60.0)
out = ref.alloc_ohlcv(idx.n - 1, True)
slab = DeviceArray(ctx, G * GiB, np.uint8)
slab.zero()
print(f"slab {slab.ptr:#x}, {G} GiB")
W = 2
def bw(off, nbytes, variant=1):
    ms, v = C.c_double(), []
    for _ in range(5):
        ctx.call("fmk_diag_read_bandwidth", C.c_void_p(slab.ptr + off), C.c_size_t(nbytes), C.c_int(variant), C.c_int(16), C.byref(ms))
        v.append(ms.value)
    return nbytes / sorted(v)[2] / 1e6
maps = []
for rnd in range(2):
    m = [bw(k * W * GiB, W * GiB) for k in range(G // W)]
    maps.append(m)
    print(f"map round {rnd} (GB/s per {W} GiB window):", " ".join(f"{x:.0f}" for x in m), flush=True)

def kernel_ms(t, reps=10, warm=4):
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

pb = (n * 8 + (2 << 20) - 1) // (2 << 20) * (2 << 20)
span = pb + n * 4
for rnd in range(2):
    for k in range(0, G - 12, 12):
        base = k * GiB
        price = DeviceArray(ctx, n, np.float64, slab.ptr + base, owner=slab)
        amount = DeviceArray(ctx, n, np.float32, slab.ptr + base + pb, owner=slab)
        ctx.call("fmk_synth_trades_dev", C.c_uint64(1), c_i64(0), c_i64(n), C.c_uint64(engine.DENSE_GAP_MOD), ref.ts.p, price.p, amount.p,
                 ref._side.p)
        t = engine.DeviceTrades(ctx, ref.ts, price, amount, None)
        ms = kernel_ms(t)
        lo, hi = k // W, (k + 12) // W
        print(f"round {rnd} columns at +{k:3d} GiB: kernel {ms:.3f} ms; map over the window {np.mean(maps[1][lo:hi]):.0f} GB/s, "
              f"read probe now {bw(base, span):.0f} GB/s", flush=True)
