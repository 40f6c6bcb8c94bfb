#!/usr/bin/env python3
"""Goldens for TimeBarReader._resample (finmlkit/bar/io.py:890-950), made by the REFERENCE's own function.

Runs only in the build container (needs /root/reference and pandas).  Inputs are 1-second bar frames built by the reference's
TimeBarKit (the caller of _resample in the reference is TimeBarReader.read on the 1-second bars AddTimeBarH5 stored,
io.py:484-485) plus frames that probe the pandas semantics the aggregation inherits: float64 volume, a second-level
resample of an already resampled frame (float32 vwap / median), NaN rows, empty seconds (zero trades), lognormal sizes
(rounding in every Kahan step), an unsorted index.  Stored: the input columns and the output frame -- data only.

    python oracle/gen_resample.py       # rewrites tests/golden/resample.npz
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
import pandas as pd  # noqa: E402

import finmlkit.bar.data_model as DM  # noqa: E402
import finmlkit.bar.kit as KIT  # noqa: E402
from finmlkit.bar.io import TimeBarReader  # noqa: E402

from oracle import oracle as orc  # noqa: E402

COLS = ["open", "high", "low", "close", "volume", "trades", "vwap", "median_trade_size"]


def one_second_bars(n, gap_mod, lognormal=False, seed=42):
    orc.build()
    ts, px, am, sd = orc.synth(seed, 0, n, gap_mod)
    if lognormal:
        am = np.random.default_rng(seed).lognormal(-1.0, 1.5, n)          # float64 sizes: inexact sums everywhere
    td = DM.TradesData(ts.copy(), px.copy(), am.copy(), np.arange(n, dtype=np.int64), side=sd.copy(), timestamp_unit="ns",
                       preprocess=False)
    return KIT.TimeBarKit(td, pd.Timedelta(seconds=1)).build_ohlcv()


def put(d, name, df, timeframe):
    res = TimeBarReader._resample(None, df, timeframe)
    d[name + "__timeframe"] = np.array(timeframe)
    d[name + "__in_index"] = df.index.values.astype("datetime64[ns]").astype(np.int64)
    for c in COLS:
        d[name + "__in_" + c] = df[c].values
    d[name + "__out_index"] = res.index.values.astype("datetime64[ns]").astype(np.int64)
    d[name + "__out_columns"] = np.array(list(res.columns))
    for c in res.columns:
        d[name + "__out_" + c] = res[c].values
    print(f"{name}: {len(df)} rows -> {len(res)} rows at {timeframe}; dtypes {dict(res.dtypes.astype(str))}")
    return res


def main():
    d = {}
    dense = one_second_bars(200_000, orc.DENSE_GAP_MOD)                    # ~10 000 seconds, ~20 ticks each
    r1 = put(d, "dense_1min", dense, "1min")
    put(d, "dense_5min", dense, "5min")
    put(d, "dense_1h", dense, "1h")
    put(d, "dense_1D", dense, "1D")
    put(d, "dense_7s", dense, "7s")
    put(d, "second_level_15min", r1, "15min")                              # float32 vwap / median in, float32 products
    sparse = one_second_bars(300, orc.SPARSE_GAP_MOD).iloc[:40_000]       # most seconds are empty: zero trades, vwap 0
    put(d, "sparse_1h", sparse, "1h")
    put(d, "sparse_1D", sparse, "1D")
    logn = one_second_bars(120_000, orc.DENSE_GAP_MOD, lognormal=True)
    put(d, "lognormal_1min", logn, "1min")
    put(d, "lognormal_30min", logn, "30min")
    f64 = logn.copy()
    f64["volume"] = f64["volume"].astype(np.float64) * 1.000001
    put(d, "f64volume_1min", f64, "1min")
    nanf = dense.iloc[:3000].copy()
    rng = np.random.default_rng(7)
    for c in ("open", "high", "low", "close", "volume", "vwap", "median_trade_size"):
        nanf.loc[nanf.index[rng.random(len(nanf)) < 0.05], c] = np.nan
    nanf.loc[nanf.index[120:180], ["open", "close"]] = np.nan              # one whole minute without an open: dropped
    put(d, "nan_1min", nanf, "1min")
    shuf = dense.iloc[:2400].iloc[np.random.default_rng(9).permutation(2400)]
    put(d, "unsorted_1min", shuf, "1min")
    path = os.path.join(ROOT, "tests", "golden", "resample.npz")
    np.savez_compressed(path, **d)
    print(f"{path}: {len(d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
