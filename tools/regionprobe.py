#!/usr/bin/env python3
"""Where inside ONE allocation of the input columns does the time-bar OHLCV + median kernel run slow?  The bars are cut into
G contiguous groups (each a contiguous range of the price / amount columns), every group is timed alone R times; then the same
for a plain read of the group's slice of the price column.  A region that is slow in every repeat is a property of where those
pages live (TLB fragments, channels); noise moves between repeats.
usage: regionprobe.py [N] [groups] [repeats] [allocations]"""
import ctypes as C, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
A = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = _ffi.default_context()
for alloc in range(A):
    gc.collect(); ctx.trim()
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
    clock, ci = t.time_bar_index(60.0)
    nb = ci.n - 1
    o = t.alloc_ohlcv(nb, True)
    cih = ci.to_host()
    for _ in range(3):
        t.bar_ohlcv(ci, True, out=o)
    ctx.sync()
    whole = []
    for _ in range(10):
        ctx.timer_start(); t.bar_ohlcv(ci, True, out=o); whole.append(ctx.timer_stop())
    print("allocation %d: price %#x amount %#x | whole step %.3f ms (min %.3f max %.3f)" % (
        alloc, t.price.ptr, t.amount.ptr, np.mean(whole), min(whole), max(whole)), flush=True)
    edges = [(g * nb) // G for g in range(G + 1)]
    tab = np.zeros((R, G)); rd = np.zeros((R, G))
    ms = C.c_double()
    for r in range(R):
        for g in range(G):
            b0, b1 = edges[g], edges[g + 1]
            v = ci.view(b0, b1 - b0 + 1)
            og = {k: a.view(b0, b1 - b0) for k, a in o.items()}
            ctx.timer_start(); t.bar_ohlcv(v, True, out=og); dt = ctx.timer_stop()
            ticks = int(cih[b1] - cih[b0])
            tab[r, g] = ticks * 12 / dt / 1e6           # GB/s of algorithmic bytes
            s0 = int(cih[b0]) + 1
            ctx.call("fmk_diag_read_bandwidth", C.c_void_p(t.price.ptr + s0 * 8), C.c_size_t(ticks * 8), C.c_int(1), C.c_int(16),
                     C.byref(ms))
            rd[r, g] = ticks * 8 / ms.value / 1e6
    med = np.median(tab, axis=0); mr = np.median(rd, axis=0)
    print("  OHLCV+median GB/s per region (median of %d): " % R + " ".join("%4.0f" % x for x in med))
    print("  spread across regions: min %.0f max %.0f (%.1f %%); repeat-to-repeat sd within a region: %.1f %%" % (
        med.min(), med.max(), 100 * (med.max() / med.min() - 1), 100 * np.mean(tab.std(axis=0) / tab.mean(axis=0))))
    print("  plain price read GB/s per region:            " + " ".join("%4.0f" % x for x in mr))
    print("  corr(region OHLCV rate, region read rate) %+.2f" % np.corrcoef(med, mr)[0, 1], flush=True)
    del t, clock, ci, o, v, og
