"""GPU parity of fmk_time_bars_ohlcv_dev (round 4): _time_bar_indexer (logic.py:12-51) + comp_bar_ohlcv (base.py:306-407) in one
call.  For streams of 1-minute-sized bars (float32 amounts) the call is PIPELINED: the clock edges of the first eighth of the bars,
then OHLCV + median of those bars while the remaining edges are searched on a second stream; both index stages take the long-bar
census, so the call never waits for a kernel it launched (csrc/fmk_ohlcv.hip).  FMK_TB_PIPE_MIN_STAGE lowers the size from which that
happens so that the small cases here take it.
Checked against the CPU oracle (clock, close indices, all eight OHLCV columns) on even, bursty, tied and degenerate timestamp
spacings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from finmlkit_amd import engine
    return engine


@pytest.fixture(autouse=True, params=["pipelined", "plain"])
def _stage_size(request, monkeypatch):
    """Every case twice: with the pipelined step from 256 bars per first stage, and with the plain two-launch order."""
    monkeypatch.setenv("FMK_TB_PIPE_MIN_STAGE", "256" if request.param == "pipelined" else "1000000000")
    yield


def _check(eng, orc, ts, px, am, interval, f64=False, what=""):
    am = np.asarray(am, np.float64 if f64 else np.float32)
    t = eng.DeviceTrades.from_numpy(ts, px, am)
    clock, idx, o = t.time_bars_ohlcv(interval)
    oclock, oci = orc._time_bar_indexer(ts, interval)
    np.testing.assert_array_equal(clock.to_host(), oclock, err_msg=f"{what}: clock")
    np.testing.assert_array_equal(idx.to_host(), oci, err_msg=f"{what}: close indices")
    want = orc.comp_bar_ohlcv(px, am, oci)
    got = eng.to_host(o)
    for k, w in zip(["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"], want):
        if k == "vwap":
            np.testing.assert_allclose(got[k], w, rtol=1e-9, err_msg=f"{what}: {k}")      # north_star: 1e-9 relative
        elif k == "volume" and f64:
            np.testing.assert_allclose(got[k], w, rtol=1e-6, err_msg=f"{what}: {k}")      # float32 of a reordered f64 sum
        else:
            np.testing.assert_array_equal(got[k], w, err_msg=f"{what}: {k}")
    # ... and the same values as the two separate calls
    c2, i2 = t.time_bar_index(interval)
    o2 = eng.to_host(t.bar_ohlcv(i2))
    np.testing.assert_array_equal(i2.to_host(), idx.to_host())
    for k in o2:
        np.testing.assert_array_equal(got[k], o2[k], err_msg=f"{what}: fused vs separate {k}")
    return len(oci) - 1


@pytest.mark.parametrize("n,interval", [(400_000, 60.0), (1_000_000, 60.0), (300_000, 45.0), (250_000, 63.7), (200_000, 35.0)])
def test_fused_matches_oracle_synth(eng, orc, n, interval):
    ts, px, am, sd = orc.synth(11, 0, n)
    nb = _check(eng, orc, ts, px, am, interval, what=f"synth n={n} I={interval}")
    assert nb >= 64 and n / nb > 600            # the fused schedule's range (shorter bars take the separate indexer, below)


@pytest.mark.parametrize("n,interval", [(200_000, 1.0), (200_000, 10.0), (100_000, 3600.0), (50_000, 86400.0), (40, 60.0)])
def test_other_schedules_through_the_same_entry(eng, orc, n, interval):
    ts, px, am, sd = orc.synth(5, 0, n)
    _check(eng, orc, ts, px, am, interval, what=f"n={n} I={interval}")


def test_float64_amounts(eng, orc):
    n = 300_000
    ts, px, am, sd = orc.synth(3, 0, n)
    am64 = np.random.default_rng(2).lognormal(-1, 1.2, n)
    _check(eng, orc, ts, px, am64, 60.0, f64=True, what="f64 amounts")


def _bursty(rng, n, mean_gap_ns):
    """Arrival times with an intraday shape and bursts: the interpolation guess is far off most of the time."""
    u = np.linspace(0, 40 * np.pi, n)
    rate = 1.0 + 0.9 * np.sin(u) + 4.0 * (rng.random(n) < 0.001)
    gaps = rng.exponential(mean_gap_ns / np.maximum(rate, 0.05)).astype(np.int64) + 1
    gaps[rng.random(n) < 0.2] = 0                                   # many equal timestamps (ties)
    gaps[0] = 1
    return 1_700_000_000_000_000_000 + np.cumsum(gaps)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_bursty_and_tied_timestamps(eng, orc, seed):
    rng = np.random.default_rng(seed)
    n = 600_000
    ts = _bursty(rng, n, 50_000_000)
    px = 100.0 + np.cumsum(rng.integers(-1, 2, n)) * 0.01
    am = rng.integers(1, 2000, n).astype(np.float32) / 64.0
    _check(eng, orc, ts, px, am, 60.0, what=f"bursty seed {seed}")


def test_gaps_empty_bars_and_edges_on_ticks(eng, orc):
    rng = np.random.default_rng(9)
    n = 500_000
    I = 60_000_000_000
    gaps = rng.integers(1, 90_000_000, n)
    gaps[rng.integers(0, n, 40)] = 7 * I + 13                       # holes of several empty bars
    ts = 1_700_000_000_000_000_000 + np.cumsum(gaps)
    # ticks exactly ON clock edges (they close the earlier bar: logic.py:42 side='right')
    e0 = (ts[0] // I) * I
    for k in (5, 17, 300, 301):
        j = np.searchsorted(ts, e0 + k * I)
        if 0 < j < n - 1:
            ts[j] = e0 + k * I
    ts = np.sort(ts)
    px = 50.0 + np.cumsum(rng.integers(-1, 2, n)) * 0.5
    am = rng.integers(1, 500, n).astype(np.float32) / 8.0
    _check(eng, orc, ts, px, am, 60.0, what="holes + ticks on edges")


def test_one_heavy_cluster(eng, orc):
    """Almost all ticks inside one minute of a long quiet day: brackets collapse only by the bisection guard."""
    rng = np.random.default_rng(4)
    n = 300_000
    quiet = np.sort(rng.integers(0, 86_400_000_000_000, 200_000))
    burst = 40_000_000_000_000 + np.sort(rng.integers(0, 50_000_000_000, n - 200_000))
    ts = 1_700_000_000_000_000_000 + np.sort(np.concatenate([quiet, burst]))
    px = 10.0 + np.cumsum(rng.integers(-1, 2, n)) * 0.25
    am = rng.integers(1, 100, n).astype(np.float32)
    _check(eng, orc, ts, px, am, 600.0, what="cluster")         # 144 bars: mean length > 600, one of them 100 000 ticks


def test_kit_build_ohlcv_uses_the_fused_call(eng, orc):
    import pandas as pd
    from finmlkit_amd.bar.data_model import TradesData
    from finmlkit_amd.bar.kit import TimeBarKit
    n = 300_000
    ts, px, am, sd = orc.synth(21, 0, n)
    tr = TradesData(ts, px, am, timestamp_unit="ns", preprocess=False)
    df = TimeBarKit(tr, pd.Timedelta(minutes=1)).build_ohlcv()
    k2 = TimeBarKit(tr, pd.Timedelta(minutes=1))
    k2._set_bar_close()                                             # the two-call path of the base class
    df2 = k2.build_ohlcv()
    pd.testing.assert_frame_equal(df, df2)
    oclock, oci = orc._time_bar_indexer(ts, 60.0)
    want = orc.comp_bar_ohlcv(px, am, oci)
    np.testing.assert_array_equal(df["trades"].values, want[6])
    np.testing.assert_array_equal(df["median_trade_size"].values, want[7])


def test_pipelined_step_with_long_bars_and_enqueue_only(eng, orc, monkeypatch):
    """The census of the index stages: a stream with a few bars beyond 1 344 ticks (the leftover passes must run), the same in
    enqueue-only mode (no read-back at all), and a stream without any (the call returns after the two launches)."""
    monkeypatch.setenv("FMK_TB_PIPE_MIN_STAGE", "256")
    rng = np.random.default_rng(12)
    n = 3_000_000
    gaps = rng.integers(1, 100_000_000, n)
    for a in (400_000, 1_700_000, 2_900_000):                       # three dense stretches: bars of ~20 000 ticks
        gaps[a:a + 60_000] = rng.integers(1, 3_000_000, 60_000)
    ts = 1_700_000_000_000_000_000 + np.cumsum(gaps)
    px = 100.0 + np.cumsum(rng.integers(-1, 2, n)) * 0.01
    am = (rng.integers(1, 4096, n) / 1024.0).astype(np.float32)
    nb = _check(eng, orc, ts, px, am, 60.0, what="long bars in a 1-minute stream")
    assert nb / 8 >= 256
    from finmlkit_amd import _ffi
    ctx = _ffi.default_context()
    ctx.set_enqueue_only(True)
    try:
        _check(eng, orc, ts, px, am, 60.0, what="long bars, enqueue-only")
    finally:
        ctx.set_enqueue_only(False)


def test_pipelined_step_at_its_default_size(eng, orc):
    """6e7 device-generated ticks (50 000 one-minute bars: the first stage has 6 144 of them without any knob): the pipelined call
    against the two separate calls on every bar, and against the oracle on the bars of the first 3e6 ticks."""
    n = 60_000_000
    t = eng.DeviceTrades.synth(n, seed=42)
    clock, idx, o = t.time_bars_ohlcv(60.0)
    c2, i2 = t.time_bar_index(60.0)
    np.testing.assert_array_equal(idx.to_host(), i2.to_host())
    np.testing.assert_array_equal(clock.to_host(), c2.to_host())
    got, o2 = eng.to_host(o), eng.to_host(t.bar_ohlcv(i2))
    for k in o2:
        np.testing.assert_array_equal(got[k], o2[k], err_msg=k)
    m = 3_000_000
    ts, px, am, sd = orc.synth(42, 0, m)
    oclock, oci = orc._time_bar_indexer(ts, 60.0)
    kb = len(oci) - 2                                               # the last bar of the prefix is cut short
    np.testing.assert_array_equal(idx.to_host()[:kb + 1], oci[:kb + 1])
    want = orc.comp_bar_ohlcv(px, am, oci[:kb + 1])
    for k, w in zip(["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"], want):
        if k == "vwap":
            np.testing.assert_allclose(got[k][:kb], w, rtol=1e-9)
        else:
            np.testing.assert_array_equal(got[k][:kb], w, err_msg=k)
