#!/usr/bin/env python3
"""Is the level of the time-bar OHLCV + median step a property of WHERE the input columns live?  K copies of the same 1e9-tick
columns are held at once (K x 21 GB) and timed round-robin, R rounds of S steps each: a level that stays with its copy through
the interleaved rounds belongs to the allocation; one that moves with time does not.  Also prints what "best of K" buys.
usage: placehold.py [N] [K] [rounds] [steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
S = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx = _ffi.default_context()
copies = [engine.DeviceTrades.synth(n, seed=42, ctx=ctx) for _ in range(K)]
clock, ci = copies[0].time_bar_index(60.0)
o = copies[0].alloc_ohlcv(ci.n - 1, True)
for t in copies:
    for _ in range(2):
        t.bar_ohlcv(ci, True, out=o)
ctx.sync()
tab = np.zeros((R, K))
for r in range(R):
    for k, t in enumerate(copies):
        ms = []
        for _ in range(S):
            ctx.timer_start(); t.bar_ohlcv(ci, True, out=o); ms.append(ctx.timer_stop())
        tab[r, k] = np.median(ms)
print("grid knob FMK_OHLCV_BLOCKS_PER_CU=%s; step ms (context timer: kernel + launch), rows = rounds, columns = copies" %
      os.environ.get("FMK_OHLCV_BLOCKS_PER_CU", "default"))
print("addresses: " + " ".join("%#x" % t.price.ptr for t in copies))
for r in range(R):
    print("  round %d: " % r + " ".join("%.3f" % x for x in tab[r]))
m = tab.mean(axis=0)
print("  mean    : " + " ".join("%.3f" % x for x in m))
within = tab.std(axis=0).mean()
between = m.std()
print("  sd within a copy across rounds %.4f ms; sd between copies %.4f ms; best %.3f  median %.3f  worst %.3f  (best/median %.3f)" % (
    within, between, m.min(), np.median(m), m.max(), m.min() / np.median(m)))
