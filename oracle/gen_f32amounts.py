#!/usr/bin/env python3
"""Reference-made vectors for the input class real data belongs to: float32 trade sizes that are NOT dyadic.

`TradesData(preprocess=True)` stores the amount column as float32 (bar/data_model.py:332-342); real sizes are decimal lots, so
every sum over them rounds.  The reference's production mode is Numba, whose typing makes `total_volume = 0.0; total_volume +=
volumes[j]` a float64 accumulation of float32 values.  The only mode that runs here is pure Python (NUMBA_DISABLE_JIT, the
reference's own CI mode), where the same line is `python float + np.float32 -> np.float32` under NEP 50 -- a float32 running
sum, NOT what production computes.  So the float32 column cannot be handed to the pure-Python reference as it is.

What this script does instead (VERDICT r2 next #2b): the amounts are drawn lognormal, rounded to float32, and handed to the
reference's own classes as `amount.astype(np.float32).astype(np.float64)` -- float64 carriers of float32 values.  Every
accumulator and every `float32 * float` product of the reducers is then float64 exactly as under Numba's promotion rules:

  comp_bar_ohlcv (base.py:306-407)            total_volume / total_dollar float64, `trade_sizes` float64 either way -> same
  comp_bar_directional_features (:409-546)    every running sum float64 -> same
  comp_bar_footprints (:615-752)              `buy_volumes[lvl] += amounts[j]`: float32 element + float64 value, rounded to
                                              float32 on the store == the float32 add Numba emits (the float64 sum of two
                                              float32 values rounds innocuously: 53 >= 2*24 + 2) -> same
  comp_footprint_features (:755-850)          NOT amount-typed: `buy_volumes[1:] * imbalance_multiplier` is float32 * python
                                              float -> float32 in this mode, float64 under Numba; `np.sum` of float32 arrays is
                                              NumPy's pairwise tree here, a sequential loop under Numba.  These two are the
                                              typed-vs-recorded divergences of DESIGN.md section 5 and do not depend on the
                                              amount column's dtype; the build follows float64 products (Numba) and the
                                              pairwise float32 sums (recorded).  The test compares the flags on all levels and
                                              reports how many differ (knife-edge levels where the float32 product rounds across
                                              the other side; none on this tape).
  comp_bar_trade_size_features (:549-612)     np.mean / np.percentile / .sum() run in the ARRAY's dtype in both modes (Numba's
                                              array reductions accumulate in the array dtype too), so this one is recorded
                                              twice: on the float32 array itself (pure-Python == NumPy float32 semantics, which
                                              the build reproduces bit for bit) and on the float64 carrier.

Inputs are not stored: ts / price / side are the first N ticks of the seed-42 stream (orc.synth), the amounts are
`default_rng(SEED).lognormal(-1, 1.2, N).astype(float32)`; tests/_golden.py:f32_amounts regenerates them.

    python oracle/gen_f32amounts.py        # ~1 min; rewrites tests/golden/f32_amounts_reference.npz
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)
os.environ["NUMBA_DISABLE_JIT"] = "1"

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

import finmlkit.bar.base as RB  # noqa: E402
import finmlkit.bar.data_model as DM  # noqa: E402
import finmlkit.bar.kit as KIT  # noqa: E402

from oracle import oracle as orc  # noqa: E402

N = 1_000_000
SEED = 777
FP_FIELDS = ["price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks", "buy_imbalances", "sell_imbalances",
             "cot_price_levels", "sell_imbalances_sum", "buy_imbalances_sum", "imb_max_run_signed", "vp_skew", "vp_gini"]


def amounts_f32(n=N, seed=SEED):
    return np.random.default_rng(seed).lognormal(-1.0, 1.2, n).astype(np.float32)


def frame(prefix, df, d):
    d[prefix + "columns"] = np.array(list(df.columns))
    for c in df.columns:
        d[prefix + "col_" + c] = df[c].values


def main():
    orc.build()
    ts, px, _, sd = orc.synth(42, 0, N)
    am32 = amounts_f32()
    am64 = am32.astype(np.float64)                       # float64 carrier of the float32 values
    assert np.array_equal(am64.astype(np.float32), am32)
    d = {"n": np.int64(N), "seed": np.int64(SEED), "amount_check": am32[::9973].copy()}
    t0 = time.time()
    td = DM.TradesData(ts.copy(), px.copy(), am64.copy(), np.arange(N, dtype=np.int64), side=sd.copy(), timestamp_unit="ns",
                       preprocess=False)
    assert td.data["amount"].dtype == np.float64
    kit = KIT.TimeBarKit(td, pd.Timedelta(minutes=1))
    ohlcv = kit.build_ohlcv()
    d["close_indices"] = np.asarray(kit._close_indices, dtype=np.int64)
    frame("ohlcv_", ohlcv, d)
    frame("dir_", kit.build_directional_features(), d)
    fp = kit.build_footprints(price_tick_size=0.01, imbalance_factor=3.0)
    nb = len(fp.bar_timestamps)
    d["fp_n_levels"] = np.array([len(fp.price_levels[i]) for i in range(nb)], dtype=np.int64)
    for k in FP_FIELDS:
        v = getattr(fp, k)
        d["fp_" + k] = v if isinstance(v, np.ndarray) and v.dtype != object else np.concatenate([np.asarray(x) for x in v])
    theta = ohlcv["median_trade_size"].values.astype(np.float64)
    d["theta"] = theta
    frame("ts64_", kit.build_trade_size_features(theta, 5.0), d)
    # the trade-size reducer on the float32 column itself (array-dtype reductions: the same in both modes)
    ci = d["close_indices"]
    for k, v in zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"],
                    RB.comp_bar_trade_size_features(am32, theta, ci, 5.0)):
        d["ts32_" + k] = np.asarray(v)
    # 1-second bars of the first 100 000 ticks: the short-bar schedules (lane per bar) on the same class of amounts
    m = 100_000
    td1 = DM.TradesData(ts[:m].copy(), px[:m].copy(), am64[:m].copy(), np.arange(m, dtype=np.int64), side=sd[:m].copy(),
                        timestamp_unit="ns", preprocess=False)
    kit1 = KIT.TimeBarKit(td1, pd.Timedelta(seconds=1))
    o1 = kit1.build_ohlcv()
    d["s1_n"] = np.int64(m)
    d["s1_close_indices"] = np.asarray(kit1._close_indices, dtype=np.int64)
    frame("s1_ohlcv_", o1, d)
    th1 = o1["median_trade_size"].values.astype(np.float64)
    d["s1_theta"] = th1
    for k, v in zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"],
                    RB.comp_bar_trade_size_features(am32[:m], th1, d["s1_close_indices"], 5.0)):
        d["s1_ts32_" + k] = np.asarray(v)
    fp1 = kit1.build_footprints(price_tick_size=0.01, imbalance_factor=3.0)
    nb1 = len(fp1.bar_timestamps)
    d["s1_fp_n_levels"] = np.array([len(fp1.price_levels[i]) for i in range(nb1)], dtype=np.int64)
    for k in FP_FIELDS:
        v = getattr(fp1, k)
        d["s1_fp_" + k] = v if isinstance(v, np.ndarray) and v.dtype != object else np.concatenate([np.asarray(x) for x in v])
    print(f"{N} ticks -> {nb} one-minute bars, {m} ticks -> {nb1} one-second bars, in {time.time() - t0:.0f} s")
    path = os.path.join(ROOT, "tests", "golden", "f32_amounts_reference.npz")
    np.savez_compressed(path, **d)
    print(f"{path}: {len(d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
