#!/usr/bin/env python3
"""Dollar-bar indexer on the synthetic tape with a share of BLOCK TRADES (sizes x `factor`): time, tier, closes against the sequential
oracle.  usage: whalebench.py [N] [share] [factor] [check: 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
share = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
factor = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
check = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
am = t.amount.to_host()
rng = np.random.default_rng(5)
idx = rng.integers(0, n, int(n * share))
am[idx] *= np.float32(factor)
t2 = engine.DeviceTrades(ctx, t.ts, t.price, DeviceArray.from_host(ctx, am), t.side)
px = t.price.to_host()
dthr = float((am[:2_000_000].astype(np.float64) * px[:2_000_000]).mean()) * 865.0 / (1 + share * factor)
os.environ["FMK_DL_VERBOSE"] = "1"
ci = t2.dollar_bar_index(dthr); ctx.sync()
os.environ["FMK_DL_VERBOSE"] = "0"
ms = []
for _ in range(3):
    ctx.sync(); s = time.perf_counter(); ci = t2.dollar_bar_index(dthr); ctx.sync(); ms.append((time.perf_counter() - s) * 1e3)
print(f"n={n:.3g} block trades {share:g} of the ticks x{factor:g}: threshold {dthr:.6g}, {ci.n - 1} closes, uncertified {t2.last_uncertified}, "
      f"{min(ms):.2f} ms (best of 3)")
if check:
    from oracle import oracle as orc
    s = time.perf_counter(); want = orc._dollar_bar_indexer(px, am, dthr); dt = time.perf_counter() - s
    got = ci.to_host()
    print(f"sequential oracle: {len(want) - 1} closes in {dt:.2f} s; equal: {np.array_equal(got, want)}")
