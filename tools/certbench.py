#!/usr/bin/env python3
"""Volume bar indexer on CONTINUOUS amounts (lognormal float64, where prefix-sum differences are not exact): time in the
default exact mode (fragile decisions on the chain are replayed) against the fast mode, per tier, and against the oracle.
With `decimal`, the amounts are tenth lots (0.1 .. 0.9) and the thresholds round numbers: exact ties on ~1/5 of the closes,
each settled by an in-kernel replay of that bar in the exact mode.
usage: certbench.py [N] [L1,L2,...] [lognormal|decimal]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from oracle import oracle as orc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
LENGTHS = [int(float(x)) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100, 865, 1500, 2500, 5000, 30000, 200000]
dist = sys.argv[3] if len(sys.argv) > 3 else "lognormal"
rng = np.random.default_rng(5)
am = rng.lognormal(0.0, 1.0, n) if dist == "lognormal" else rng.integers(1, 10, n) / 10.0
ctx = _ffi.default_context()
t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), np.ones(n), am, ctx=ctx)
mean_v = float(am.mean())
print(f"n={n} {dist} float64 amounts, mean {mean_v:.4f}")
for L in LENGTHS:
    thr = mean_v * L if dist == "lognormal" else float(round(mean_v * L))
    row = []
    res = {}
    for mode in ("exact", "fast"):
        ctx.set_fast_threshold(mode == "fast")
        ci = t.volume_bar_index(thr); ctx.sync()
        ms = []
        for _ in range(3):
            ctx.timer_start(); ci = t.volume_bar_index(thr); ms.append(ctx.timer_stop())
        res[mode] = ci.to_host()
        row.append("%s %8.2f ms (%d uncertified)" % (mode, min(ms), t.last_uncertified))
    t0 = time.time(); want = orc._volume_bar_indexer(am, thr); t1 = time.time()
    ok = np.array_equal(res["exact"], want)
    print("mean bar %7d ticks: %s   %d bars, exact == oracle: %s, fast == oracle: %s   (oracle %.1f s)" %
          (L, "   ".join(row), len(want) - 1, ok, np.array_equal(res["fast"], want), t1 - t0), flush=True)
ctx.set_fast_threshold(False)
