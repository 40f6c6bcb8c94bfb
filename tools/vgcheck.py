#!/usr/bin/env python3
"""Volume bar indexer against the oracle on CONTINUOUS amounts and long bars (the global-table tier of fmk_volume.hip):
lognormal float64 / float32 amounts, bar lengths 3 000 .. 40 000, heavy tails, short streams, near-tie thresholds.
usage: vgcheck.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd.bar import logic
from oracle import oracle as orc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
bad = 0
for seed, sigma, dtype, L in [(1, 1.0, np.float64, 3500), (2, 1.0, np.float64, 5000), (3, 0.5, np.float32, 8000),
                              (4, 2.5, np.float64, 6000), (5, 1.0, np.float64, 20000), (6, 3.0, np.float64, 12000),
                              (7, 1.0, np.float64, 40000), (8, 0.1, np.float64, 3100), (9, 1.0, np.float32, 3300)]:
    rng = np.random.default_rng(seed)
    for nn in (n, 70_001, 5_000):
        am = rng.lognormal(0.0, sigma, nn).astype(dtype)
        if seed == 6:
            am[rng.integers(0, nn, 20)] *= 5e4          # whales: bars of one tick among the long ones
        thr = float(am.astype(np.float64).mean()) * L
        for t in (thr, float(np.cumsum(am.astype(np.float64))[min(nn - 1, L)])):      # the second one is a knife edge for bar 1
            t0 = time.time(); got = logic._volume_bar_indexer(am, t); t1 = time.time()
            want = orc._volume_bar_indexer(am, t)
            ok = np.array_equal(np.asarray(got), np.asarray(want))
            bad += not ok
            print(f"seed {seed} sigma {sigma} {np.dtype(dtype).name} n={nn} L~{L} thr={t:.6g}: {len(want) - 1} bars "
                  f"{'OK' if ok else 'MISMATCH'} ({(t1 - t0) * 1e3:.0f} ms)", flush=True)
            if not ok:
                g, w = np.asarray(got), np.asarray(want)
                k = int(np.argmax(g[:min(len(g), len(w))] != w[:min(len(g), len(w))])) if len(g) and len(w) else 0
                print("   lens", len(g), len(w), "first diff at", k, g[max(0, k - 1):k + 3], w[max(0, k - 1):k + 3])
print("mismatches:", bad)
sys.exit(1 if bad else 0)
