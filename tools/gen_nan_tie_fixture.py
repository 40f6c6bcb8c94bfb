#!/usr/bin/env python3
"""tests/golden/nan_tie_longbar.npz: the inputs of tools/fuzz_longbars.py seed 361, case 70 (five bars, one of 65 537 ticks whose running signed
dollar sum peaks 2.7e-12 below a float32 rounding boundary at tick 488 and holds a NaN amount at tick 554), compactly: price steps as int8,
amounts as multiples of 2^-10 in uint16 (0 = the NaN), sides int8, and the REFERENCE's own outputs (comp_bar_directional_features of
/root/reference in pure-Python mode through oracle/shim, float64 carriers of the float32 sizes as in oracle/gen_longbars.py; the oracle must
agree with them, or this script stops).  Build container only.  Replays the fuzz tool's generator on the CPU."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import fuzz_longbars as F
from oracle import oracle as orc
from tests import _golden as G
orc.build()
cap = {}
class Fake:
    def comp_bar_ohlcv(self, px, am, ci): return orc.comp_bar_ohlcv(px, am, ci)
    def comp_bar_directional_features(self, px, am, ci, sd):
        cap["last"] = (px.copy(), am.copy(), ci.copy(), sd.copy())
        return orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    def comp_bar_footprints_csr(self, *a): return orc.comp_bar_footprints_csr(*a)
    def comp_bar_trade_size_features(self, *a): return orc.comp_bar_trade_size_features(*a)
rng = np.random.default_rng(361)
for k in range(71):
    F.case(rng, orc, {"base": Fake()}, k, False)
px, am, ci, sd = cap["last"]
step = 0.05
walk = np.rint((px - 100.0) / step).astype(np.int64)
steps = np.diff(np.concatenate([[0], walk]))
assert np.abs(steps).max() <= 2
px2 = np.maximum(100.0 + step * np.cumsum(steps), step)
assert np.array_equal(px2, px)
units = np.where(np.isnan(am), 0, np.rint(am * 1024.0)).astype(np.int64)
assert units.max() < 65536 and units.min() >= 0
am2 = (units * 2.0 ** -10).astype(np.float32); am2[units == 0] = np.nan
assert np.array_equal(am2, am, equal_nan=True)
sys.path.insert(0, os.path.join(ROOT, "oracle", "shim"))
sys.path.insert(1, "/root/reference")
os.environ["NUMBA_DISABLE_JIT"] = "1"
import finmlkit.bar.base as RB  # noqa: E402
ref = RB.comp_bar_directional_features(px, am.astype(np.float64), ci, sd)
want = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
for k, r, w in zip(G.DIR_KEYS, ref, want):
    r = np.asarray(r)
    assert np.array_equal(r.astype(w.dtype), w, equal_nan=True), (k, r, w)
out = {f"want_{k}": v for k, v in zip(G.DIR_KEYS, want)}
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nan_tie_longbar.npz"), steps=steps.astype(np.int8), units=units.astype(np.uint16),
                    sides=sd.astype(np.int8), close_idx=ci, **out)
print("written", len(px), "ticks, bars", np.diff(ci))
