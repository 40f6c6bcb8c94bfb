#!/usr/bin/env python3
"""For rocprofv3 --pmc: map a 144 GiB slab with the dominant kernel, then launch it 4x on the fastest and 4x on the slowest 2 GiB window (in
that order: the LAST eight dispatches of k_bar_ohlcv_small in the counter file).  Prints the two windows and their kernel times."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
G = 144
n = 150_000_000
ctx = _ffi.default_context()
GiB = 1 << 30
ref = engine.DeviceTrades.synth(n, seed=1, first=0, ctx=ctx)
clock, idx = ref.time_bar_index(60.0)
out = ref.alloc_ohlcv(idx.n - 1, True)
slab = DeviceArray(ctx, G * GiB, np.uint8)
slab.zero()
A_OFF = 1280 << 20
def trades_at(w, fill=True):
    base = w * 2 * GiB
    price = DeviceArray(ctx, n, np.float64, slab.ptr + base, owner=slab)
    amount = DeviceArray(ctx, n, np.float32, slab.ptr + base + A_OFF, owner=slab)
    if fill:
        ctx.call("fmk_synth_trades_dev", C.c_uint64(1), c_i64(0), c_i64(n), C.c_uint64(engine.DENSE_GAP_MOD), ref.ts.p, price.p, amount.p,
                 ref._side.p)
    return engine.DeviceTrades(ctx, ref.ts, price, amount, None)
def kernel_ms(t, reps=3, warm=1):
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
m = [kernel_ms(trades_at(k)) for k in range(G // 2)]
print("kernel us per 2 GiB window:", " ".join(f"{x * 1e3:.0f}" for x in m), flush=True)
order = np.argsort(m)
fast, slow = int(order[1]), int(order[-2])
print(f"fast window {fast} ({m[fast]*1e3:.0f} us), slow window {slow} ({m[slow]*1e3:.0f} us)")
for w in (fast, slow):
    t = trades_at(w, fill=False)
    for _ in range(4):
        t.bar_ohlcv(idx, want_median=True, out=out)
    ctx.sync()
