#!/usr/bin/env python3
"""Fine map of one large slab with the dominant kernel itself: 1.5e8 ticks (price 1.2 GB + amount 0.6 GB) placed in every 2 GiB window,
kernel time per window, two rounds; then the same with the two columns 1 GiB windows apart (price in window k, amount in window k+1).
usage: placemap2.py [slab GiB]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
G = int(sys.argv[1]) if len(sys.argv) > 1 else 144
n = 150_000_000
ctx = _ffi.default_context()
GiB = 1 << 30
ref = engine.DeviceTrades.synth(n, seed=1, first=0, ctx=ctx)
clock, idx = ref.time_bar_index(60.0)
out = ref.alloc_ohlcv(idx.n - 1, True)
slab = DeviceArray(ctx, G * GiB, np.uint8)
slab.zero()
print(f"slab {slab.ptr:#x}, {G} GiB; {n} ticks, {idx.n - 1} bars per window")

def kernel_ms(t, reps=8, warm=3):
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

def at(p_off, a_off):
    price = DeviceArray(ctx, n, np.float64, slab.ptr + p_off, owner=slab)
    amount = DeviceArray(ctx, n, np.float32, slab.ptr + a_off, owner=slab)
    ctx.call("fmk_synth_trades_dev", C.c_uint64(1), c_i64(0), c_i64(n), C.c_uint64(engine.DENSE_GAP_MOD), ref.ts.p, price.p, amount.p,
             ref._side.p)
    return kernel_ms(engine.DeviceTrades(ctx, ref.ts, price, amount, None))

print("reference allocation:", round(kernel_ms(ref), 4), "ms")
W = 2
for rnd in range(2):
    m = [at(k * W * GiB, k * W * GiB + 1280 * (1 << 20)) for k in range(G // W)]
    print(f"round {rnd} kernel us per {W} GiB window:", " ".join(f"{x * 1e3:.0f}" for x in m), flush=True)
# price alone in a slow / fast window, amount in a fast / slow one
order = np.argsort(m)
fast, slow = int(order[0]), int(order[-1])
print(f"fast window {fast} ({m[fast]*1e3:.0f} us), slow window {slow} ({m[slow]*1e3:.0f} us)")
for pw, aw, what in ((fast, fast, "both fast"), (slow, slow, "both slow"), (slow, fast, "price slow, amount fast"), (fast, slow, "price fast, amount slow")):
    print(f"{what}: {at(pw * W * GiB, aw * W * GiB + 1280 * (1 << 20)) * 1e3:.0f} us")
