#!/usr/bin/env python3
"""Largest relative deviation of the device's ewmst / ewmst_mean0 from the sequential C oracle on 1e6 synthetic ticks, for a
range of half lives (also through tools/ab_lib.py against another build).  usage: ewdev.py [half_life ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd.feature.core.volatility import ewmst, ewmst_mean0
from oracle import oracle as orc
hls = [float(x) for x in sys.argv[1:]] or [0.05, 0.5, 5.0, 60.0, 600.0]
ts, px, am, sd = orc.synth(33, 0, 1_000_000)
r = orc.comp_lagged_returns(ts, px, 2.0, True)
r[5000:5040] = np.nan
for hl in hls:
    for name, fn, ofn in (("ewmst", ewmst, orc.ewmst), ("ewmst_mean0", ewmst_mean0, orc.ewmst_mean0)):
        got, want = fn(ts, r, hl), ofn(ts, r, hl)
        ok = np.isfinite(want) & (want != 0)
        rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        i = int(np.argmax(rel))
        print(f"half_life {hl:g} {name}: max rel {rel.max():.2e} (at {np.flatnonzero(ok)[i]}: {got[ok][i]!r} vs {want[ok][i]!r}), "
              f"99.9 % quantile {np.quantile(rel, 0.999):.2e}, NaN pattern equal {np.array_equal(np.isnan(got), np.isnan(want))}, "
              f"zeros equal {np.array_equal(got[want == 0], want[want == 0])}", flush=True)
