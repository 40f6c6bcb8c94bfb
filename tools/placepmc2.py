#!/usr/bin/env python3
"""Placement, with counters (time-boxed): the SAME step on the slowest and on the fastest of K positions of the input columns
inside one slab, in one process -- first every position is timed (HIP events), then the slowest and the fastest are synthesised again and the step
runs `reps` times on each, bracketed by marker kernels: marker(1) slow... marker(2) fast... marker(1).  Under rocprofv3 --pmc the counters of the
dominant kernel's dispatches in the two regions are the comparison (tools/placepmc2.sh prints it).   usage: placepmc2.py [ticks] [positions] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
K = int(sys.argv[2]) if len(sys.argv) > 2 else 13
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = _ffi.default_context()
span = (21 * n + (8 << 20) + (1 << 30) - 1) // (1 << 30) * (1 << 30)
step = 8 << 30
slab = DeviceArray(ctx, (K - 1) * step + span, np.uint8)


def timed(t, k=6):
    for _ in range(3):
        t.time_bars_ohlcv(60.0)
    best = 1e9
    for _ in range(k):
        ctx.timer_start(); t.time_bars_ohlcv(60.0); best = min(best, ctx.timer_stop())
    return best


ms = []
for i in range(K):
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx, into=(slab, i * step))
    ms.append(timed(t))
    del t
slow, fast = int(np.argmax(ms)), int(np.argmin(ms))
print("PLACE step ms by position (every 8 GiB): " + " ".join("%.3f" % m for m in ms), flush=True)
print(f"PLACE slow position {slow} ({ms[slow]:.3f} ms), fast position {fast} ({ms[fast]:.3f} ms)", flush=True)
for which, pos in ((1, slow), (2, fast)):
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx, into=(slab, pos * step))
    for _ in range(3):
        t.time_bars_ohlcv(60.0)
    ctx.sync()
    ctx.call("fmk_diag_marker_dev", C.c_int(which))
    for _ in range(reps):
        t.time_bars_ohlcv(60.0)
    ctx.sync()
    del t
ctx.call("fmk_diag_marker_dev", C.c_int(1))
ctx.sync()
