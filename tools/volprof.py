#!/usr/bin/env python3
"""volume_bar_index at N ticks, cfg-3 threshold (mean bar 865 ticks), default exact mode: 3 timed calls (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
L = float(sys.argv[2]) if len(sys.argv) > 2 else 864.6
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
probe = engine.DeviceTrades.synth(1_000_000, seed=42, ctx=ctx)
thr = float(probe.amount.to_host().astype(np.float64).mean()) * L
for _ in range(4):
    ctx.sync(); t0 = time.perf_counter(); ci = t.volume_bar_index(thr); ctx.sync()
    print(f"n={n:.3g} L={L}: {ci.n - 1} bars, {1e3 * (time.perf_counter() - t0):.2f} ms, uncertified {t.last_uncertified}", flush=True)
