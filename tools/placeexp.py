#!/usr/bin/env python3
"""Where do the +-5 % between allocations of the input columns come from?  One slab per trial, the price and amount columns synthesised
INTO it at chosen byte offsets; the dominant kernel (comp_bar_ohlcv + median over 1-minute bars, close indices computed once) timed by
the library's HIP events.  usage: placeexp.py [ticks] [mode]   mode: offsets | slabs | shift"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
mode = sys.argv[2] if len(sys.argv) > 2 else "offsets"
ctx = _ffi.default_context()
MiB = 1 << 20

ref = engine.DeviceTrades.synth(n, seed=1, first=0, ctx=ctx)
clock, idx = ref.time_bar_index(60.0)
nb = idx.n - 1
out = ref.alloc_ohlcv(nb, True)
ts = ref.ts
side = ref._side


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
    return v[len(v) // 2], v[0]


def place(slab, p_off, a_off):
    """price at slab + p_off, amount at slab + a_off (bytes); the synthetic stream written there"""
    price = DeviceArray(ctx, n, np.float64, slab.ptr + p_off, owner=slab)
    amount = DeviceArray(ctx, n, np.float32, slab.ptr + a_off, owner=slab)
    ctx.call("fmk_synth_trades_dev", C.c_uint64(1), c_i64(0), c_i64(n), C.c_uint64(engine.DENSE_GAP_MOD), ts.p, price.p, amount.p,
             side.p)
    return engine.DeviceTrades(ctx, ts, price, amount, None)


print(f"reference allocation (price {ref.price.ptr:#x}, amount {ref.amount.ptr:#x}): median / min kernel ms", *kernel_ms(ref))
pb = (n * 8 + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)            # price column rounded up to 2 MiB
if mode == "offsets":
    slab = DeviceArray(ctx, pb + n * 4 + 64 * MiB, np.uint8)
    print(f"slab {slab.ptr:#x}")
    for rnd in range(2):
        for d in (0, 256, 1024, 4096, 16384, 65536, 262144, MiB, 2 * MiB + 4096, 8 * MiB, 32 * MiB):
            t = place(slab, 0, pb + d)
            m, lo = kernel_ms(t)
            print(f"round {rnd} amount at price_end + {d:>9d} B: median {m:.3f} min {lo:.3f} ms", flush=True)
elif mode == "slabs":
    # six slabs held at once, the same relative layout in each, two rounds: does the level belong to the slab?
    slabs = [DeviceArray(ctx, pb + n * 4 + 64 * MiB, np.uint8) for _ in range(6)]
    ts_ = [place(s, 0, pb) for s in slabs]
    for rnd in range(2):
        for k, t in enumerate(ts_):
            m, lo = kernel_ms(t)
            print(f"round {rnd} slab {k} at {slabs[k].ptr:#x}: median {m:.3f} min {lo:.3f} ms", flush=True)
elif mode == "shift":
    # ONE slab four times the size, the pair of columns at different positions in it
    span = pb + n * 4 + 64 * MiB
    slab = DeviceArray(ctx, 4 * span + 1024 * MiB, np.uint8)
    print(f"slab {slab.ptr:#x}")
    for rnd in range(2):
        for k in range(4):
            for extra in (0, 2 * MiB, 512 * MiB):
                base = k * span + extra
                t = place(slab, base, base + pb)
                m, lo = kernel_ms(t)
                print(f"round {rnd} columns at slab + {base / 2**30:8.3f} GiB: median {m:.3f} min {lo:.3f} ms", flush=True)
