#!/usr/bin/env python3
"""What do the slow regions of device memory (tools/placemap2.py) slow down?  The slab is mapped with the dominant kernel (2 GiB windows), then
the fastest and the slowest window are read by the two-stream probe (8 B + 4 B elements in lock-step) in three patterns and at several
occupancies, and by the flat read probe.  usage: placeprobe.py [slab GiB]"""
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
A_OFF = 1280 << 20

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

def at(w):
    base = w * 2 * GiB
    price = DeviceArray(ctx, n, np.float64, slab.ptr + base, owner=slab)
    amount = DeviceArray(ctx, n, np.float32, slab.ptr + base + A_OFF, owner=slab)
    ctx.call("fmk_synth_trades_dev", C.c_uint64(1), c_i64(0), c_i64(n), C.c_uint64(engine.DENSE_GAP_MOD), ref.ts.p, price.p, amount.p,
             ref._side.p)
    return kernel_ms(engine.DeviceTrades(ctx, ref.ts, price, amount, None))

m = [at(k) for k in range(G // 2)]
print("kernel us per 2 GiB window:", " ".join(f"{x * 1e3:.0f}" for x in m), flush=True)
order = np.argsort(m)
fast, slow = int(order[1]), int(order[-2])
print(f"fast window {fast} ({m[fast]*1e3:.0f} us), slow window {slow} ({m[slow]*1e3:.0f} us)")

def two(w, pattern, seg, bpc):
    ms, v = C.c_double(), []
    base = slab.ptr + w * 2 * GiB
    for _ in range(7):
        ctx.call("fmk_diag_read_two_streams", C.c_void_p(base), C.c_void_p(base + A_OFF), C.c_int64(n), C.c_int(pattern), C.c_int(seg),
                 C.c_int(bpc), C.byref(ms))
        v.append(ms.value)
    return sorted(v)[3] * 1e3

def flat(w, variant, bpc):
    ms, v = C.c_double(), []
    for _ in range(7):
        ctx.call("fmk_diag_read_bandwidth", C.c_void_p(slab.ptr + w * 2 * GiB), C.c_size_t(n * 12), C.c_int(variant), C.c_int(bpc), C.byref(ms))
        v.append(ms.value)
    return sorted(v)[3] * 1e3

print("probe (us, fast window / slow window / ratio): segments that are / are not a whole number of 128-byte lines")
for pattern in (3, 1):
    for seg in (1200, 1203, 1237, 1191):
        for bpc in (2, 4, 16):
            a, b = two(fast, pattern, seg, bpc), two(slow, pattern, seg, bpc)
            print(f"  two streams, pattern {pattern}, seg {seg:4d}, {bpc:2d} blocks/CU: {a:7.1f} {b:7.1f} {b / a:.3f}", flush=True)
