"""GPU: the HIP path fed the float32 NON-dyadic amount column against vectors the REFERENCE made (VERDICT r2 next #2b).

The vectors (tests/golden/f32_amounts_reference.npz, oracle/gen_f32amounts.py) come from the reference's own TimeBarKit run on
float64 carriers of the float32 values, which makes every accumulator of the reducers float64 as under Numba's typing -- the
production semantics this build follows.  Both call routes are covered: the separate reducers (build_ohlcv /
build_directional_features / build_footprints / build_trade_size_features) and the fused cfg-4 pass, on one-minute bars (wave
and workgroup schedules) and on one-second bars (lane-per-bar schedules)."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu


def _run(orc, prefix, ci_key, n_key, interval, fused):
    from finmlkit_amd import engine
    d = G.load("f32_amounts_reference")
    n = int(d[n_key])
    ts, px, _, sd = orc.synth(42, 0, n)
    am = G.f32_amounts(d)[:n]
    t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
    assert not t.amount_is_f64
    clock, ci = t.time_bar_index(interval)
    np.testing.assert_array_equal(ci.to_host(), d[ci_key])
    if fused:
        o, dd, nz, off, flat, bar, bad = t.bars_fused(ci, 0.01, 3.0, want_median=True)
    else:
        o = t.bar_ohlcv(ci)
        dd, nz = t.bar_directional(ci)
        off, flat, bar, bad = t.bar_footprints(ci, o["low"], o["high"], 0.01, 3.0)
    assert int(bad.to_host()[0]) == 0
    o, flat, bar = engine.to_host(o), engine.to_host(flat), engine.to_host(bar)
    dd = engine.to_host(dd) if prefix == "" else None          # one-second bars: empty seconds -> mean_spread undefined
    theta = d[prefix + "theta"] if prefix else d["theta"]
    np.testing.assert_array_equal(o["median_trade_size"], theta)
    t32 = t.bar_trade_size(ci, theta, 5.0)
    nd = G.check_f32_amount_vectors(d, prefix, n, ci_key, o, dd, (np.diff(off.to_host()), flat, bar), t32,
                                    what=f"HIP {'fused' if fused else 'separate'} {prefix or '1min'}")
    assert nd <= 2, nd
    return t, ci, theta, d


@pytest.mark.parametrize("fused", [False, True])
def test_one_minute_bars_float32_amounts_vs_reference(orc, fused):
    t, ci, theta, d = _run(orc, "", "close_indices", "n", 60.0, fused)
    if not fused:
        # the float64 carrier through the trade-size reducer as well (what the reference run itself was given)
        from finmlkit_amd import engine
        ts, px, am, sd = t.to_numpy()
        t64 = engine.DeviceTrades.from_numpy(ts, px, am.astype(np.float64), sd)
        got = t64.bar_trade_size(ci, theta, 5.0)
        for key in ("mean_size_rel", "size_95_rel", "pct_block", "size_gini"):
            np.testing.assert_array_equal(got[key], d["ts64_col_" + key], err_msg=key)


@pytest.mark.parametrize("fused", [False, True])
def test_one_second_bars_float32_amounts_vs_reference(orc, fused):
    _run(orc, "s1_", "s1_close_indices", "s1_n", 1.0, fused)


@pytest.mark.parametrize("fused", [False, True])
def test_long_bars_float32_amounts_vs_reference(orc, fused):
    """Bars of 70 001 / 100 / 129 900 / 1 / 16 499 / 8 500 / 194 999 ticks (tests/golden/long_bars_reference.npz, made by
    oracle/gen_longbars.py with the reference's own four reducers): the workgroup-per-bar schedules -- OHLCV + median, order flow
    with its tick-order redo, footprints on one LDS histogram, trade-size features by chunks of np.sum -- against the REFERENCE."""
    from finmlkit_amd import engine
    d = G.load("long_bars_reference")
    n = int(d["lb_n"])
    ts, px, _, sd = orc.synth(42, 0, n)
    am = G.long_bars_amounts()
    t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
    assert not t.amount_is_f64
    ci = engine.DeviceArray.from_host(t.ctx, d["lb_close_indices"])
    if fused:
        o, dd, nz, off, flat, bar, bad = t.bars_fused(ci, 0.01, 3.0, want_median=True)
    else:
        o = t.bar_ohlcv(ci)
        dd, nz = t.bar_directional(ci)
        off, flat, bar, bad = t.bar_footprints(ci, o["low"], o["high"], 0.01, 3.0)
    assert int(bad.to_host()[0]) == 0
    o, flat, bar, dd = engine.to_host(o), engine.to_host(flat), engine.to_host(bar), engine.to_host(dd)
    theta = d["lb_theta"]
    np.testing.assert_array_equal(o["median_trade_size"], theta)
    t32 = t.bar_trade_size(ci, theta, 5.0)
    nd = G.check_f32_amount_vectors(d, "lb_", n, "lb_close_indices", o, dd, (np.diff(off.to_host()), flat, bar), t32,
                                    what=f"HIP {'fused' if fused else 'separate'} long bars")
    assert nd <= 2, nd
