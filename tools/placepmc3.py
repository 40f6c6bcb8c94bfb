#!/usr/bin/env python3
"""Placement with counters, cheap form: the time-bar step on K positions (16 GiB apart) of one slab, `reps` steps each, a marker kernel between positions.
Under `rocprofv3 --kernel-trace --pmc ...` every dispatch of the dominant kernel has its duration AND its counters: tools/placepmc3.sh correlates them.
usage: placepmc3.py [ticks] [positions] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = _ffi.default_context()
span = (21 * n + (8 << 20) + (1 << 30) - 1) // (1 << 30) * (1 << 30)
step = 16 << 30
slab = DeviceArray(ctx, (K - 1) * step + span, np.uint8)
for i in range(K):
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx, into=(slab, i * step))
    t.time_bars_ohlcv(60.0)                      # (buffers, first touch)
    ctx.sync()
    ctx.call("fmk_diag_marker_dev", C.c_int(i + 1))
    for _ in range(reps):
        t.time_bars_ohlcv(60.0)
    ctx.sync()
    del t
ctx.call("fmk_diag_marker_dev", C.c_int(0))
ctx.sync()
