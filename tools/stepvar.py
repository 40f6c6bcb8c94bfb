#!/usr/bin/env python3
"""What the bench step adds to the dominant kernel's time.  K copies of the columns held at once; per copy, S launches each of
  (a) OHLCV + median alone, (b) time-bar index then OHLCV + median (the bench step), both read through the library's own HIP-event
  pair around the dominant launch (fmk_profile_*), and the wall time per step.  usage: stepvar.py [N] [K] [S]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = _ffi.default_context()
copies = [engine.DeviceTrades.synth(n, seed=42, ctx=ctx) for _ in range(K)]
clock, ci = copies[0].time_bar_index(60.0)
ne = ci.n
o = copies[0].alloc_ohlcv(ne - 1, True)
bufs = (DeviceArray(ctx, ne + 1024, np.int64), DeviceArray(ctx, ne + 1024, np.int64))


def prof(fn):
    ctx.sync()
    ctx.call("fmk_profile_enable", C.c_int(1))
    t0 = time.perf_counter()
    for _ in range(S):
        fn()
    ctx.sync()
    wall = (time.perf_counter() - t0) / S * 1e3
    kms = (C.c_double * 64)(); kn = C.c_int()
    ctx.call("fmk_profile_read", kms, C.c_int(64), C.byref(kn))
    ctx.call("fmk_profile_enable", C.c_int(0))
    k = np.array([kms[i] for i in range(kn.value)])
    return k, wall


print("grid knob FMK_OHLCV_BLOCKS_PER_CU=%s" % os.environ.get("FMK_OHLCV_BLOCKS_PER_CU", "default"))
for rnd in range(2):
    for k, t in enumerate(copies):
        for _ in range(3):
            t.bar_ohlcv(ci, True, out=o)
        a, wa = prof(lambda: t.bar_ohlcv(ci, True, out=o))
        def step():
            c2, i2 = t.time_bar_index(60.0, out=bufs)
            t.bar_ohlcv(i2, True, out=o)
        step()
        b, wb = prof(step)
        print("round %d copy %d: kernel alone  mean %.3f min %.3f max %.3f (wall/step %.3f) | after the indexer  mean %.3f min %.3f max %.3f "
              "(wall/step %.3f)" % (rnd, k, a.mean(), a.min(), a.max(), wa, b.mean(), b.min(), b.max(), wb), flush=True)
