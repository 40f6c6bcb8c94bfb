"""GPU: the reference's QuickStart flow end to end through the drop-in API -- raw exchange rows ->
TradesData(preprocess=True) -> time / tick / volume / dollar / CUSUM kits -> OHLCV, order-flow, footprints,
trade-size features -> rolling volume profile -> ReturnT | EWMST transforms.  Every stage is compared with the CPU
oracle fed with the same (preprocessed) arrays."""
import numpy as np
import pandas as pd
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu


def _raw_rows(orc, n):
    ts, px, am, sd = orc.synth(77, 0, n)
    rng = np.random.default_rng(0)
    # exchange-style rows: millisecond timestamps (=> same-timestamp prints), ids in arrival order, shuffled
    ts_ms = ts // 1_000_000
    ids = np.arange(n, dtype=np.int64) + 10
    maker = sd == -1                                  # taker sold <=> buyer was the maker
    perm = rng.permutation(n)
    return ts_ms[perm], px[perm], am.astype(np.float64)[perm], ids[perm], maker[perm], perm


def test_quickstart_flow(orc):
    from finmlkit_amd.bar.data_model import TradesData
    from finmlkit_amd.bar.kit import CUSUMBarKit, DollarBarKit, TickBarKit, TimeBarKit, VolumeBarKit
    from finmlkit_amd.feature.core.volume import VolumePro
    from finmlkit_amd.feature.transforms import EWMST, Compose, ReturnT
    n = 400_000
    ts_ms, px, qty, ids, maker, perm = _raw_rows(orc, n)
    # the reference applies `is_buyer_maker` as given (not re-ordered with the id sort): hand it over in id order
    maker_sorted = maker[np.argsort(ids, kind="stable")]
    trades = TradesData(ts_ms, px, qty, ids, is_buyer_maker=maker_sorted, preprocess=True, name="quickstart")
    df = trades.data
    assert list(df.columns) == ["timestamp", "price", "amount", "side"] and df["amount"].dtype == np.float32
    assert trades.data_ok and len(df) < n and df["timestamp"].is_monotonic_increasing
    ts, p, a, s = (df[c].values for c in ("timestamp", "price", "amount", "side"))

    # ---- time bars: all four builders
    kit = TimeBarKit(trades, pd.Timedelta(minutes=1))
    ohlcv = kit.build_ohlcv()
    clock, ci = orc._time_bar_indexer(ts, 60.0)
    want = orc.comp_bar_ohlcv(p, a, ci)
    np.testing.assert_array_equal(kit.bar_close_indices, ci[1:])
    for col, w in zip(["open", "high", "low", "close", "volume"], want[:5]):
        np.testing.assert_array_equal(ohlcv[col].values, w, err_msg=col)
    G.assert_f64_close(ohlcv["vwap"].values, want[5], rtol=1e-9, what="vwap")
    np.testing.assert_array_equal(ohlcv["median_trade_size"].values, want[7])
    dirf = kit.build_directional_features()
    wd = orc.comp_bar_directional_features(p, a, ci, s)
    np.testing.assert_array_equal(dirf["ticks_buy"].values, wd[0])
    np.testing.assert_array_equal(dirf["volume_buy"].values, wd[2], err_msg="volume_buy")
    fp = kit.build_footprints(price_tick_size=0.01)
    woff, wflat, wbar = orc.comp_bar_footprints_csr(p, a, ci, s, 0.01, want[2], want[1], 3.0)
    np.testing.assert_array_equal(fp.level_offsets, woff)
    np.testing.assert_array_equal(fp.flat["buy_volumes"], wflat["buy_volumes"])
    np.testing.assert_array_equal(fp.cot_price_levels, wbar["cot_price_levels"])
    theta = np.full(len(ohlcv), float(np.median(a)))
    tsf = kit.build_trade_size_features(theta)
    wts = orc.comp_bar_trade_size_features(a, theta, ci, 5.0)
    np.testing.assert_allclose(tsf["mean_size_rel"].values, wts[0], rtol=2e-5)

    # ---- rolling volume profile on the footprints
    poc, hva, lva, pct = VolumePro(pd.Timedelta(minutes=30), n_bins=27).compute(ohlcv, fp)
    wv = orc.volume_profile_rolling(fp.bar_timestamps, want[1], want[2], woff, wflat["price_levels"],
                                    wflat["buy_volumes"], wflat["sell_volumes"], 1800.0, 27, 0.01, 68.34)
    np.testing.assert_array_equal(np.nan_to_num(poc), wv[0] * 0.01)
    np.testing.assert_array_equal(pct, wv[3])

    # ---- the other bar types
    np.testing.assert_array_equal(TickBarKit(trades, 500).bar_close_indices, orc._tick_bar_indexer(ts, 500)[1:])
    vthr = float(a.sum()) / 300
    np.testing.assert_array_equal(VolumeBarKit(trades, vthr).bar_close_indices, orc._volume_bar_indexer(a, vthr)[1:])
    dthr = float((p * a).sum()) / 300
    np.testing.assert_array_equal(DollarBarKit(trades, dthr).bar_close_indices, orc._dollar_bar_indexer(p, a, dthr)[1:])

    # ---- tick-level volatility -> CUSUM bars
    sigma = Compose(ReturnT(pd.Timedelta(seconds=5), is_log=True, input_col="price"),
                    EWMST(pd.Timedelta(seconds=60)))(df).values
    wr = orc.comp_lagged_returns(ts, p, 5.0, True)
    G.assert_f64_close(sigma, orc.ewmst(ts, wr, 60.0), rtol=1e-9, what="sigma")
    ck = CUSUMBarKit(trades, sigma.copy(), sigma_floor=2e-5, sigma_mult=2.0)
    cus = ck.build_ohlcv()
    wci = orc._cusum_bar_indexer(ts, p, orc.ewmst(ts, wr, 60.0), 2e-5, 2.0)
    assert len(cus) > 20
    # sigma differs from the oracle's by ~1e-15 relative (scan order), so a close can move only at an exact tie
    np.testing.assert_array_equal(ck.bar_close_indices, wci[1:])
