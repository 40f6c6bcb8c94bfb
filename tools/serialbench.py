#!/usr/bin/env python3
"""ns per tick of the exact serial loop (k_threshold_exact: the default-mode fallback of the dollar indexer, and of the
volume indexer when a replayed decision disagrees).  usage: serialbench.py [N]"""
import os, sys
os.environ["FMK_THRESHOLD_SERIAL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from oracle import oracle as orc
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
rng = np.random.default_rng(3)
am = rng.lognormal(0, 1, n)
px = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=n)), 0.01)
ctx = _ffi.default_context()
for dt, L in ((np.float64, 50), (np.float32, 50), (np.float64, 3), (np.float64, 1000)):
    a = am.astype(dt)
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), px, a, ctx=ctx)
    for kind, thr, fn, want in (("volume", float(a.mean()) * L, t.volume_bar_index, lambda th: orc._volume_bar_indexer(a, th)),
                                ("dollar", float((a.astype(np.float64) * px).mean()) * L, t.dollar_bar_index,
                                 lambda th: orc._dollar_bar_indexer(px, a, th))):
        got = fn(thr).to_host(); ctx.sync()
        ms = []
        for _ in range(2):
            ctx.timer_start(); r = fn(thr); ms.append(ctx.timer_stop())
        print(f"{kind:6s} {np.dtype(dt).name} bars of ~{L} ticks: {min(ms):8.1f} ms for {n} ticks = {min(ms) * 1e6 / n:5.1f} ns/tick, "
              f"{len(got) - 1} bars, equal to the oracle: {np.array_equal(got, want(thr))}", flush=True)
