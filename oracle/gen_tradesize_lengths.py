#!/usr/bin/env python3
"""Reference-made vectors for comp_bar_trade_size_features (finmlkit/bar/base.py:549-612) over the BAR-LENGTH axis -- test
infrastructure, runs in the build container only (imports /root/reference in pure-Python mode through oracle/shim).

The reducer's NumPy reductions (np.mean, np.percentile, .sum() of a float32 slice) run in the array's dtype, so the reference in
pure-Python mode computes what the jitted reference computes for float32 amounts.  The HIP path serves a
bar by one of seven schedules chosen by its length (one lane, sixteen lanes, one wave reading the bar once, 2 / 4 / 8 / 16 waves,
a workgroup with the tree cut into sub-trees); this fixture holds the reference's outputs for bars on both sides of every edge
between them, for three kinds of float32 sizes (lognormal: all distinct; decimal lots: heavy ties; dyadic), with a NaN size, a zero
theta and an all-zero bar among them.  Inputs are regenerated from the seed (tests/_golden.py: tradesize_lengths_inputs); stored:
the close indices, theta and the four output columns per kind.
    python oracle/gen_tradesize_lengths.py      -> tests/golden/trade_size_lengths_reference.npz   (~10 s)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)
os.environ["NUMBA_DISABLE_JIT"] = "1"

import numpy as np  # noqa: E402

import finmlkit.bar.base as RB  # noqa: E402

from tests._golden import tradesize_lengths_inputs, TS_LENGTH_KINDS  # noqa: E402


def main():
    d = {}
    for kind in TS_LENGTH_KINDS:
        am, theta, ci = tradesize_lengths_inputs(kind)
        with np.errstate(all="ignore"):
            out = RB.comp_bar_trade_size_features(am, theta, ci, 5.0)
        d[kind + "_close_indices"] = ci
        d[kind + "_theta"] = theta
        d[kind + "_amount_check"] = am[::997].copy()
        for k, v in zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"], out):
            d[kind + "_" + k] = np.asarray(v)
        print(kind, len(ci) - 1, "bars,", len(am), "ticks; NaN rows:", int(np.isnan(np.asarray(out[0])).sum()))
    path = os.path.join(ROOT, "tests", "golden", "trade_size_lengths_reference.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
