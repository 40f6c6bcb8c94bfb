#!/usr/bin/env python3
"""comp_bar_ohlcv on SHORT bars: N ticks -> time bars of the given intervals (1 s = the reference's other caller,
AddTimeBarH5, bar/io.py:484-485: ~20 ticks per bar on the synthetic stream), with and without the median.
usage: shortbars.py [N] [interval,interval,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ivs = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1.0, 2.0, 5.0, 10.0, 60.0]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
for iv in ivs:
    clock, ci = t.time_bar_index(iv)
    nb = ci.n - 1
    for med in (True, False):
        out = t.alloc_ohlcv(nb, med)
        t.bar_ohlcv(ci, want_median=med, out=out); ctx.sync()
        ms = []
        for _ in range(5):
            ctx.timer_start(); t.bar_ohlcv(ci, want_median=med, out=out); ms.append(ctx.timer_stop())
        best = min(ms)
        gb = (12.0 * n + (68 if med else 60) * nb + 8 * (nb + 1)) / 1e9
        print(f"n={n:.3g} interval {iv:g} s: {nb} bars ({n / nb:.1f} ticks/bar) median={med}: {best:.3f} ms  "
              f"{n / best / 1e6:.1f} Gticks/s  {gb / best * 1e3:.0f} GB/s algorithmic", flush=True)
        del out
    del clock, ci
