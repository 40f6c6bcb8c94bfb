#!/usr/bin/env python3
"""The pipelined time-bar step against the two separate calls on ONE allocation of the inputs, alternating blocks of S steps:
kernel time per step (sum of the dominant kernel's launches, HIP events) and wall time per step.  usage: pipeab.py [N] [S] [rounds]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
clock, ci = t.time_bar_index(60.0)
ne = ci.n
o = t.alloc_ohlcv(ne + 1024, True)
bufs = (DeviceArray(ctx, ne + 1024, np.int64), DeviceArray(ctx, ne + 1024, np.int64))


def prof(fn):
    ctx.sync()
    ctx.call("fmk_profile_enable", C.c_int(1))
    t0 = time.perf_counter()
    for _ in range(S):
        fn()
    ctx.sync()
    wall = (time.perf_counter() - t0) / S * 1e3
    kms = (C.c_double * 256)(); kn = C.c_int()
    ctx.call("fmk_profile_read", kms, C.c_int(256), C.byref(kn))
    ctx.call("fmk_profile_enable", C.c_int(0))
    return sum(kms[i] for i in range(kn.value)) / S, wall


def piped():
    t.time_bars_ohlcv(60.0, True, out_index=bufs, out=o)


def separate():
    c2, i2 = t.time_bar_index(60.0, out=bufs)
    t.bar_ohlcv(i2, True, out=o)


for _ in range(10):
    piped(); separate()
for r in range(R):
    kp, wp = prof(piped)
    ks, ws = prof(separate)
    print("round %d: pipelined kernel %.3f wall %.3f (diff %.3f) | separate kernel %.3f wall %.3f (diff %.3f)" % (r, kp, wp, wp - kp, ks, ws, ws - ks), flush=True)
