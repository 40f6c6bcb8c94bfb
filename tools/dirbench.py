#!/usr/bin/env python3
"""comp_bar_directional_features alone at N ticks, 1-minute bars (and other intervals): FMK_DIR_LANES=0 wave per bar, default /
2 one lane per bar.  usage: dirbench.py [N] [interval_s ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(x) for x in sys.argv[2:]] or [60.0]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    for mode in ("0", "1", "2"):
        os.environ["FMK_DIR_LANES"] = mode
        ms = []
        for _ in range(4):
            ctx.timer_start(); d, nz = t.bar_directional(ci); ms.append(ctx.timer_stop()); del d, nz
        print(f"n={n:.3g} interval {iv:g}s ({ci.n - 1} bars): FMK_DIR_LANES={mode}: {min(ms):.3f} ms", flush=True)
