#!/usr/bin/env python3
"""The placement study of round 4 in one script (its siblings -- allocwarm, arenawarm, placeexp, placemap, placeprobe, placeshift, placepmc
... -- were removed in round 6; they are in the repository's history, their results in profiles/r04_placement*.txt and
profiles/r05_placement_counters.txt).  Is the allocation-dependent step time a property of the memory system or of the OHLCV kernel?
One process, R rounds: free everything, give the pooled blocks back to the driver, allocate + fill the input columns, then
on the SAME allocation time (a) the plain read probe over the price column (8 B/lane and 16 B/lane variants, 8 GB) and
(b) 20 steps of time-bar OHLCV + median.  Prints both per round and their correlation.
usage: placement.py [N] [rounds]"""
import ctypes as C, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = _ffi.default_context()


def probe(arr, variant):
    ms, best = C.c_double(), []
    for _ in range(6):
        ctx.call("fmk_diag_read_bandwidth", arr.p, C.c_size_t(arr.n * arr.dtype.itemsize), C.c_int(variant), C.c_int(16),
                 C.byref(ms))
        best.append(ms.value)
    return arr.n * arr.dtype.itemsize / np.median(best[1:]) / 1e6      # GB/s


def probe2(t, pattern):
    ms, out = C.c_double(), []
    for _ in range(6):
        ctx.call("fmk_diag_read_two_streams", t.price.p, t.amount.p, C.c_int64(t.n), C.c_int(pattern), C.c_int(1200),
                 C.c_int(16), C.byref(ms))
        out.append(ms.value)
    return t.n * (8 if pattern == 2 else 12) / np.median(out[1:]) / 1e6


rows = []
for r in range(rounds):
    gc.collect(); ctx.trim()
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
    clock, ci = t.time_bar_index(60.0)
    o = t.alloc_ohlcv(ci.n - 1, True)
    for _ in range(3):
        t.bar_ohlcv(ci, True, out=o)
    ctx.sync()
    ms = []
    for _ in range(20):
        ctx.timer_start(); t.bar_ohlcv(ci, True, out=o); ms.append(ctx.timer_stop())
    p8, pa = probe(t.price, 1), probe(t.amount, 1)
    two = [probe2(t, pat) for pat in (0, 1, 2)]
    rows.append((float(np.mean(ms)), p8, pa, *two))
    print("round %2d: OHLCV step %.3f ms | GB/s: price alone %5.0f  amount alone %5.0f | price+amount flat %5.0f  "
          "bar-walk %5.0f | price bar-walk %5.0f" % (r, *rows[-1]), flush=True)
    del t, clock, ci, o
a = np.array(rows)
names = ["OHLCV step ms", "price alone", "amount alone", "price+amount flat", "price+amount bar-walk", "price bar-walk"]
for k, nm in enumerate(names):
    print("%-24s min %8.3f  max %8.3f  spread %5.1f %%%s" % (nm, a[:, k].min(), a[:, k].max(), 100 * (a[:, k].max() / a[:, k].min() - 1),
          "" if k == 0 else "   corr(step time, 1/bandwidth) %+.2f" % np.corrcoef(a[:, 0], 1 / a[:, k])[0, 1]))
