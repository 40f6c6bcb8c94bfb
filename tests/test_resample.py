"""TimeBarReader._resample (finmlkit/bar/io.py:890-950): the C oracle (CPU) and the HIP path (GPU) against frames the
reference's own function produced (oracle/gen_resample.py -> tests/golden/resample.npz).  Every column bit-exact: the sums
are pandas' Kahan recurrences in the column's dtype, nothing is reassociated."""
import numpy as np
import pandas as pd
import pytest

from tests import _golden as G

COLS = ["open", "high", "low", "close", "volume", "trades", "vwap", "median_trade_size"]
CASES = ["dense_1min", "dense_5min", "dense_1h", "dense_1D", "dense_7s", "second_level_15min", "sparse_1h", "sparse_1D",
         "lognormal_1min", "lognormal_30min", "f64volume_1min", "nan_1min", "unsorted_1min"]


def _frame(d, case):
    df = pd.DataFrame({c: d[f"{case}__in_{c}"] for c in COLS},
                      index=pd.DatetimeIndex(d[f"{case}__in_index"].astype("datetime64[ns]"), name="timestamp"))
    return df, str(d[f"{case}__timeframe"])


def _check(d, case, got: pd.DataFrame):
    assert list(got.columns) == [str(c) for c in d[f"{case}__out_columns"]] == COLS
    np.testing.assert_array_equal(got.index.values.astype("datetime64[ns]").astype(np.int64), d[f"{case}__out_index"])
    for c in COLS:
        want = d[f"{case}__out_{c}"]
        assert got[c].values.dtype == want.dtype, (case, c, got[c].values.dtype, want.dtype)
        np.testing.assert_array_equal(got[c].values, want, err_msg=f"{case}:{c}")


@pytest.mark.parametrize("case", CASES)
def test_oracle_resample_golden(orc, case):
    d = G.load("resample")
    df, timeframe = _frame(d, case)
    codes, uniques = pd.factorize(df.index.floor(timeframe), sort=False)
    order = np.argsort(codes, kind="stable")
    codes = codes[order]
    seg = np.concatenate([[0], np.flatnonzero(np.diff(codes)) + 1, [len(df)]]).astype(np.int64)
    out = orc.resample_bars(seg, *[df[c].values[order] for c in COLS])
    got = pd.DataFrame(dict(zip(COLS, out[:8])), index=uniques)[out[8].astype(bool)]
    _check(d, case, got)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_resample_golden(case):
    from finmlkit_amd.bar.io import TimeBarReader
    d = G.load("resample")
    df, timeframe = _frame(d, case)
    _check(d, case, TimeBarReader()._resample(df, timeframe))


@pytest.mark.gpu
def test_hip_resample_one_second_bars_of_a_long_stream(orc):
    """The reference's other caller builds 1-SECOND bars (AddTimeBarH5, io.py:484-485) and TimeBarReader resamples them: 3e6
    ticks -> ~150 000 one-second bars (HIP) -> 1-minute / 1-hour / 1-day bars (HIP) against the oracle on the same frame; and
    the resampled 1-minute OHLC / volume / trades against 1-minute bars built directly from the ticks."""
    from finmlkit_amd import engine
    from finmlkit_amd.bar.io import resample_bars
    n = 3_000_000
    t = engine.DeviceTrades.synth(n, seed=42)
    clock, ci = t.time_bar_index(1.0)
    o = engine.to_host(t.bar_ohlcv(ci))
    idx = pd.DatetimeIndex(clock.to_host()[1:].astype("datetime64[ns]"), name="timestamp")
    df = pd.DataFrame({"open": o["open"], "high": o["high"], "low": o["low"], "close": o["close"], "volume": o["volume"],
                       "trades": o["trades"], "vwap": o["vwap"], "median_trade_size": o["median_trade_size"]}, index=idx)
    for timeframe in ("1min", "1h", "1D"):
        got = resample_bars(df, timeframe)
        codes, uniques = pd.factorize(df.index.floor(timeframe), sort=False)
        seg = np.concatenate([[0], np.flatnonzero(np.diff(codes)) + 1, [len(df)]]).astype(np.int64)
        want = orc.resample_bars(seg, *[df[c].values for c in COLS])
        assert len(got) == len(uniques)
        for c, w in zip(COLS, want[:8]):
            np.testing.assert_array_equal(got[c].values, w, err_msg=f"{timeframe}:{c}")
    # 1-second bars close on (edge, edge + 1 s]; floor() puts the bar labelled with an exact minute edge into the NEXT minute,
    # so the two aggregations agree on everything that does not involve that one bar: compare the totals
    got = resample_bars(df, "1min")
    assert int(got["trades"].sum()) == int(o["trades"].sum())
    assert float(got["high"].max()) == float(o["high"].max()) and float(got["low"].min()) == float(o["low"].min())
