#!/usr/bin/env python3
"""BASELINE.json configs[0] ("cfg 1"): 10M synthetic ticks -> 1-minute time bars through the REFERENCE itself.

Runs only in the build container (needs /root/reference; pure-Python mode as the reference's CI pins it, see
oracle/gen_golden.py).  The first 10^7 ticks of the seed-42 stream of SURVEY.md 8(d) -- the prefix of the 10^9-tick
stream bench.py and tests/test_gpu_fullsize.py run -- go through the reference's own classes:

    TradesData(ts, price, amount, id, side=..., timestamp_unit="ns", preprocess=False)
    TimeBarKit(trades, pd.Timedelta(minutes=1)).build_ohlcv()                 (bar/kit.py:12-35, bar/base.py:132-169)

and, on the first 10^6 ticks, .build_directional_features() and .build_footprints(price_tick_size=0.01).  Stored: the
kit's close timestamps / close indices and every output column (8 333 bars, < 1 MB) -- data only.  The oracle is checked
against it on the CPU (tests/test_oracle_golden.py), the 10^9-tick HIP run's first bars directly on the GPU
(tests/test_gpu_fullsize.py): this ties the headline configuration to vectors the reference made, not only to the oracle.

The stream's amounts are float32 multiples of 2^-10 whose per-bar sums are exact in float32, so the reference's
pure-Python accumulation (float32 under NEP 50) and its Numba-typed one (float64) are the same numbers.

(The tick-level chain lagged returns -> ewmst -> CUSUM has a generator of its own, oracle/gen_ticklevel_chain.py: the reference's
pure-Python loops need ~10 minutes per 10^6 ticks there, and a generator nobody re-runs stops being a pin.)

    python oracle/gen_cfg1.py            # ~3 min; rewrites tests/golden/cfg1_reference_timebars.npz
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

import finmlkit.bar.data_model as DM  # noqa: E402
import finmlkit.bar.kit as KIT  # noqa: E402

from oracle import oracle as orc  # noqa: E402

N_OHLCV = 10_000_000
N_FLOW = 1_000_000
FP_FIELDS = ["bar_timestamps", "price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks", "buy_imbalances",
             "sell_imbalances", "cot_price_levels", "sell_imbalances_sum", "buy_imbalances_sum", "imb_max_run_signed",
             "vp_skew", "vp_gini"]


def kit_for(n):
    orc.build()
    ts, px, am, sd = orc.synth(42, 0, n)
    td = DM.TradesData(ts.copy(), px.copy(), am.copy(), np.arange(n, dtype=np.int64), side=sd.copy(),
                       timestamp_unit="ns", preprocess=False)
    assert td.data["amount"].dtype == np.float32 and td.data["price"].dtype == np.float64
    assert np.array_equal(td.data["timestamp"].values, ts)
    return KIT.TimeBarKit(td, pd.Timedelta(minutes=1))


def frame(prefix, df, d):
    d[prefix + "index_ns"] = df.index.values.astype("datetime64[ns]").astype(np.int64)
    d[prefix + "columns"] = np.array(list(df.columns))
    for c in df.columns:
        d[prefix + "col_" + c] = df[c].values


def main():
    d = {"n_ohlcv": np.int64(N_OHLCV), "n_flow": np.int64(N_FLOW), "seed": np.int64(42)}
    t0 = time.time()
    kit = kit_for(N_OHLCV)
    ohlcv = kit.build_ohlcv()
    d["close_ts"] = np.asarray(kit._close_ts, dtype=np.int64)
    d["close_indices"] = np.asarray(kit._close_indices, dtype=np.int64)
    frame("ohlcv_", ohlcv, d)
    print(f"build_ohlcv: {N_OHLCV} ticks -> {len(ohlcv)} bars in {time.time() - t0:.0f} s")

    t0 = time.time()
    kit = kit_for(N_FLOW)
    frame("flow_ohlcv_", kit.build_ohlcv(), d)
    d["flow_close_indices"] = np.asarray(kit._close_indices, dtype=np.int64)
    frame("dir_", kit.build_directional_features(), d)
    fp = kit.build_footprints(price_tick_size=0.01, imbalance_factor=3.0)
    nb = len(fp.bar_timestamps)
    d["fp_n_levels"] = np.array([len(fp.price_levels[i]) for i in range(nb)], dtype=np.int64)
    for k in FP_FIELDS:
        v = getattr(fp, k)
        if isinstance(v, np.ndarray) and v.dtype != object:
            d["fp_" + k] = v
        else:
            d["fp_" + k] = np.concatenate([np.asarray(x) for x in v]) if nb else np.zeros(0)
    d["fp_price_tick"] = np.float64(fp.price_tick)
    print(f"directional + footprints: {N_FLOW} ticks -> {nb} bars in {time.time() - t0:.0f} s")

    # ---- cfg 3 at the same size: the reference's own sequential indexers (bar/logic.py:87-149) on the first 10^7 ticks with
    #      the thresholds the bench derives from the data (about one bar per 865 ticks), and on a lognormal float64 tape where
    #      every addition rounds.  Only the close indices are stored.
    import finmlkit.bar.logic as LG
    t0 = time.time()
    orc.build()
    ts, px, am, sd = orc.synth(42, 0, N_OHLCV)
    vthr, dthr = 1728.5, 17285000.0
    d["cfg3_vthr"], d["cfg3_dthr"] = np.float64(vthr), np.float64(dthr)
    d["cfg3_volume_close_indices"] = np.array(LG._volume_bar_indexer(am, vthr), dtype=np.int64)
    d["cfg3_dollar_close_indices"] = np.array(LG._dollar_bar_indexer(px, am, dthr), dtype=np.int64)
    rng = np.random.default_rng(2024)
    m = 2_000_000
    lam = rng.lognormal(-1.0, 1.2, m)
    lpx = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=m)), 0.01)
    d["cfg3_logn_seed"], d["cfg3_logn_n"] = np.int64(2024), np.int64(m)
    lv, ld = float(np.mean(lam)) * 700.0, float(np.mean(lam * lpx)) * 700.0
    d["cfg3_logn_vthr"], d["cfg3_logn_dthr"] = np.float64(lv), np.float64(ld)
    d["cfg3_logn_volume_close_indices"] = np.array(LG._volume_bar_indexer(lam, lv), dtype=np.int64)
    d["cfg3_logn_dollar_close_indices"] = np.array(LG._dollar_bar_indexer(lpx, lam, ld), dtype=np.int64)
    print(f"volume / dollar indexers: {len(d['cfg3_volume_close_indices']) - 1} / {len(d['cfg3_dollar_close_indices']) - 1} bars on "
          f"{N_OHLCV} ticks, {len(d['cfg3_logn_volume_close_indices']) - 1} / {len(d['cfg3_logn_dollar_close_indices']) - 1} on the "
          f"lognormal tape, in {time.time() - t0:.0f} s")

    path = os.path.join(ROOT, "tests", "golden", "cfg1_reference_timebars.npz")
    np.savez_compressed(path, **d)
    print(f"{path}: {len(d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
