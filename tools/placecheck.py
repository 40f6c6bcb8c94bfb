#!/usr/bin/env python3
"""Does the level a 5-step probe reads for an allocation of the input columns hold in the timed region that follows (bench.py's
choose_placement)?  K copies; three probe rounds over all of them; then, like the bench, the others are freed, fresh per-bar
buffers are made and the chosen copy runs three blocks of 10 steps.  usage: placecheck.py [N] [K] [free: 0|1] [newbuf: 0|1]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
do_free = int(sys.argv[3]) if len(sys.argv) > 3 else 1
new_buf = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ctx = _ffi.default_context()
copies = [engine.DeviceTrades.synth(n, seed=42, ctx=ctx) for _ in range(K)]


def prof(fn, S):
    ctx.sync()
    ctx.call("fmk_profile_enable", C.c_int(1))
    for _ in range(S):
        fn()
    ctx.sync()
    kms = (C.c_double * 64)(); kn = C.c_int()
    ctx.call("fmk_profile_read", kms, C.c_int(64), C.byref(kn))
    ctx.call("fmk_profile_enable", C.c_int(0))
    k = np.array([kms[i] for i in range(kn.value)])
    return k


clock = idx = out = None
for t in copies:
    for _ in range(2):
        clock, idx, out = t.time_bars_ohlcv(60.0, True, out_index=(clock, idx) if clock else None, out=out)
lvl = np.zeros((3, K))
for rnd in range(3):
    for k, t in enumerate(copies):
        a = prof(lambda: t.time_bars_ohlcv(60.0, True, out_index=(clock, idx), out=out), 5)
        lvl[rnd, k] = a.mean()
    print("probe round %d: %s" % (rnd, "  ".join("%.3f" % x for x in lvl[rnd])), flush=True)
best = int(np.argmin(lvl[0]))
print("chosen by round 0:", best)
if do_free:
    for i, t in enumerate(copies):
        if i != best:
            for col in t._backing:
                col.free()
t = copies[best]
if new_buf:
    ne = idx.n
    clock = DeviceArray(ctx, ne + 1024, np.int64); idx = DeviceArray(ctx, ne + 1024, np.int64); out = t.alloc_ohlcv(ne + 1024, True)
for blk in range(3):
    a = prof(lambda: t.time_bars_ohlcv(60.0, True, out_index=(clock, idx), out=out), 10)
    print("timed block %d on copy %d (free %d, new buffers %d): mean %.3f min %.3f max %.3f" % (blk, best, do_free, new_buf, a.mean(), a.min(), a.max()), flush=True)
