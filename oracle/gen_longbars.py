#!/usr/bin/env python3
"""Reference-made vectors for the four bar reducers on LONG bars -- test infrastructure, runs in the build container only (imports
/root/reference in pure-Python mode through oracle/shim).

Every fixture made before this one holds bars of ~1 200 ticks (one-minute bars of the synthetic tape) or shorter; the schedules that
serve long bars in the HIP path (a workgroup per bar: k_bar_ohlcv_wide / _mid / _phased, k_bar_dir_wide and its tick-order redo,
k_bar_footprints_wide, k_bar_trade_size_wg / _wide) were pinned to the ORACLE only -- and the oracle had been wrong about np.sum
beyond 8 192 elements.  Here the reference itself runs comp_bar_ohlcv, comp_bar_directional_features,
comp_bar_footprints and comp_bar_trade_size_features on 420 000 ticks cut into bars of 70 001 / 100 / 129 900 / 1 / 16 499 / 8 500 /
194 999 ticks, float32 lognormal sizes (as float64 carriers for the three functions whose scalar accumulators are float64 under
Numba's typing -- oracle/gen_f32amounts.py explains --, as the float32 array itself for the trade-size reducer).  Key layout of
tests/_golden.py: check_f32_amount_vectors with prefix "lb_".
    python oracle/gen_longbars.py      -> tests/golden/long_bars_reference.npz   (~2 min)
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

import finmlkit.bar.base as RB  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from tests._golden import LONG_BARS_N, LONG_BARS_CUTS, long_bars_amounts, FP_LIST_KEYS  # noqa: E402

OHLCV = ["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"]
DIRC = ["ticks_buy", "ticks_sell", "volume_buy", "volume_sell", "dollars_buy", "dollars_sell", "mean_spread", "max_spread",
        "cum_ticks_min", "cum_ticks_max", "cum_volume_min", "cum_volume_max", "cum_dollars_min", "cum_dollars_max"]     # base.py:538-546


def main():
    orc.build()
    n = LONG_BARS_N
    ts, px, _, sd = orc.synth(42, 0, n)
    am32 = long_bars_amounts()
    am64 = am32.astype(np.float64)
    ci = np.array(LONG_BARS_CUTS, dtype=np.int64)
    d = {"lb_n": np.int64(n), "lb_close_indices": ci, "lb_amount_check": am32[::9973].copy()}
    t0 = time.time()
    o = RB.comp_bar_ohlcv(px, am64, ci)
    d["lb_ohlcv_columns"] = np.array(OHLCV)
    for k, v in zip(OHLCV, o):
        d["lb_ohlcv_col_" + k] = np.asarray(v)
    print("ohlcv", round(time.time() - t0), "s", flush=True)
    dd = RB.comp_bar_directional_features(px, am64, ci, sd)
    d["lb_dir_columns"] = np.array(DIRC)
    for k, v in zip(DIRC, dd):
        d["lb_dir_col_" + k] = np.asarray(v)
    print("directional", round(time.time() - t0), "s", flush=True)
    lows, highs = np.asarray(o[2]), np.asarray(o[1])
    fp = RB.comp_bar_footprints(px, am64, ci, sd, 0.01, lows, highs, 3.0)
    names = ["price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks", "buy_imbalances", "sell_imbalances",
             "buy_imbalances_sum", "sell_imbalances_sum", "cot_price_levels", "imb_max_run_signed", "vp_skew", "vp_gini"]
    nb = len(ci) - 1
    d["lb_fp_n_levels"] = np.array([len(fp[0][i]) for i in range(nb)], dtype=np.int64)
    for k, v in zip(names, fp):
        d["lb_fp_" + k] = np.asarray(v) if k not in names[:7] else np.concatenate([np.asarray(x) for x in v])
    print("footprints", round(time.time() - t0), "s; levels per bar", d["lb_fp_n_levels"], flush=True)
    theta = np.asarray(o[7]).astype(np.float64)
    d["lb_theta"] = theta
    for k, v in zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"], RB.comp_bar_trade_size_features(am32, theta, ci, 5.0)):
        d["lb_ts32_" + k] = np.asarray(v)
    path = os.path.join(ROOT, "tests", "golden", "long_bars_reference.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes in", round(time.time() - t0), "s")


if __name__ == "__main__":
    main()
