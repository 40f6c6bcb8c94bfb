#!/usr/bin/env python3
"""Time-bar indexer alone (fmk_time_bar_indexer_dev) on 1e9 resident ticks, context timer, best / median of R calls per interval.
FMK_TIME_INDEX_INTERP=0 selects the two-level search with the sample table (rounds 1-3).  usage: indexbench.py [N] [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
print("FMK_TIME_INDEX_INTERP=%s" % os.environ.get("FMK_TIME_INDEX_INTERP", "default(1)"))
for iv in (60.0, 1.0, 10.0, 600.0, 3600.0):
    clock, idx = t.time_bar_index(iv)
    ms = []
    for _ in range(R):
        ctx.timer_start(); t.time_bar_index(iv, out=(clock, idx)); ms.append(ctx.timer_stop())
    print("interval %7.1f s: %8d edges  best %.3f ms  median %.3f ms" % (iv, idx.n, min(ms), float(np.median(ms))), flush=True)
