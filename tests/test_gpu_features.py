"""GPU parity: comp_bar_directional_features and comp_bar_footprints (+ comp_footprint_features)
vs the CPU oracle and the reference-generated golden fixtures, through the C ABI."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu

INT_DIR = {"ticks_buy", "ticks_sell", "cum_ticks_min", "cum_ticks_max"}


def _check_dir(got, want, what):
    for k, g, w in zip(G.DIR_KEYS, got, want):
        assert g.dtype == w.dtype, (what, k)
        # float32 columns too: bars whose float64 sums sit within rounding noise of a float32 tie are redone in the
        # reference's tick order on the device, so every output is bit-identical
        np.testing.assert_array_equal(g, w, err_msg=f"{what}:{k}")


@pytest.mark.parametrize("case", ["syn_t60", "syn_t1", "syn_tick100", "syn_vol2048", "rnd_t120", "rnd_tick37"])
def test_directional_golden(orc, case):
    from finmlkit_amd.bar.base import comp_bar_directional_features
    d = G.load("reducers")
    px, am, sd = G.reducer_stream(orc, d, case)
    got = comp_bar_directional_features(px, am, d[f"{case}__ci"], sd)
    _check_dir(got, tuple(d[f"{case}__dir_{k}"] for k in G.DIR_KEYS), case)


@pytest.mark.parametrize("n,interval,f64,zeros", [(300_000, 60.0, False, False), (200_000, 1.0, True, True),
                                                  (300_000, 7200.0, True, False)])
def test_directional_vs_oracle(orc, n, interval, f64, zeros):
    from finmlkit_amd.bar.base import comp_bar_directional_features
    ts, px, am, sd = orc.synth(5, 0, n)
    rng = np.random.default_rng(9)
    if f64:
        am = rng.lognormal(-1, 1.3, n)
    if zeros:
        sd = sd.copy()
        sd[rng.random(n) < 0.1] = 0
    _, ci = orc._time_bar_indexer(ts, interval)
    want = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    empty = np.diff(ci) <= 0
    if np.isnan(want[6]).any():
        with pytest.raises(ZeroDivisionError):
            comp_bar_directional_features(px, am, ci, sd)
    else:
        _check_dir(comp_bar_directional_features(px, am, ci, sd), want, f"n={n} iv={interval}")


@pytest.mark.parametrize("case", ["t60", "t1", "t7200", "volume", "sparse", "zeros", "tail"])
def test_directional_lane_per_bar_schedule(orc, monkeypatch, case):
    """k_bar_dir_lanes (one lane walks one bar in tick order; forced here, the library picks it for >= 64 K moderate bars):
    non-dyadic float32 amounts, 20-tick to 9 000-tick bars in one call (bars beyond 8 192 ticks go to the wave-per-bar kernel
    through the list), empty bars, unsigned ticks before a bar's first signed one, a stream that ends inside a 16-tick block."""
    from finmlkit_amd import _ffi, engine
    monkeypatch.setenv("FMK_DIR_LANES", "2")
    n = 700_003 if case == "tail" else 600_000
    gap = 400_000_000_000 if case == "sparse" else None
    ts, px, am, sd = orc.synth(23, 0, n, gap) if gap else orc.synth(23, 0, n)
    rng = np.random.default_rng(4)
    am = rng.lognormal(-1, 1.3, n).astype(np.float32)
    if case == "zeros":
        sd = sd.copy()
        sd[rng.random(n) < 0.3] = 0
        sd[:40] = 0
    if case == "volume":
        ci = orc._volume_bar_indexer(am.astype(np.float64), float(am.mean()) * 300.0)
        ci = np.concatenate([ci[:200], ci[260:]])                  # one bar of ~18 000 ticks among ~300-tick bars
    else:
        _, ci = orc._time_bar_indexer(ts, {"t1": 1.0, "t7200": 7200.0}.get(case, 60.0))
    want = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
    d, nz = t.bar_directional(_ffi.DeviceArray.from_host(t.ctx, ci))
    got = tuple(d[k].to_host() for k in G.DIR_KEYS)
    assert int(nz.to_host()[0]) == int(np.isnan(want[6]).sum())
    for k, g, w in zip(G.DIR_KEYS, got, want):
        np.testing.assert_array_equal(g, w, err_msg=f"{case}:{k}")


def test_directional_zero_division(orc):
    from finmlkit_amd.bar.base import comp_bar_directional_features
    ts, px, am, sd = orc.synth(42, 0, 5_000, 500_000_000_000)       # sparse stream: empty bars
    _, ci = orc._time_bar_indexer(ts, 60.0)
    with pytest.raises(ZeroDivisionError):
        comp_bar_directional_features(px, am, ci, sd)


def _check_fp(off, flat, bar, woff, wflat, wbar, what):
    np.testing.assert_array_equal(off, woff, err_msg=what)
    for k in G.FP_LIST_KEYS:
        np.testing.assert_array_equal(flat[k].astype(wflat[k].dtype), wflat[k], err_msg=f"{what}:{k}")
    for k in G.FP_BAR_KEYS:
        if k == "vp_skew":    # identically 0 in exact arithmetic: rounding noise of the reference's dot product
            np.testing.assert_allclose(bar[k], wbar[k], rtol=0, atol=1e-6, err_msg=f"{what}:{k}")
        else:
            np.testing.assert_array_equal(bar[k], wbar[k], err_msg=f"{what}:{k}")


@pytest.mark.parametrize("case", ["syn_t60", "syn_t1", "syn_tick100", "syn_vol2048", "rnd_t120", "rnd_tick37"])
def test_footprints_golden(orc, case):
    from finmlkit_amd.bar.base import comp_bar_footprints_csr
    d = G.load("reducers")
    px, am, sd = G.reducer_stream(orc, d, case)
    off, flat, bar = comp_bar_footprints_csr(px, am, d[f"{case}__ci"], sd, 0.01, d[f"{case}__ohlcv_low"],
                                             d[f"{case}__ohlcv_high"], 3.0)
    _check_fp(off, flat, bar, d[f"{case}__fp_offsets"], {k: d[f"{case}__fp_{k}"] for k in G.FP_LIST_KEYS},
              {k: d[f"{case}__fp_{k}"] for k in G.FP_BAR_KEYS}, case)


@pytest.mark.parametrize("n,interval,dtype,tick,mult", [(300_000, 60.0, np.float32, 0.01, 3.0),
                                                        (300_000, 3600.0, np.float32, 0.01, 1.5),
                                                        (150_000, 10.0, np.float64, 0.01, 2.0),
                                                        (100_000, 900.0, np.float32, 0.002, 3.0)])
def test_footprints_vs_oracle(orc, n, interval, dtype, tick, mult):
    """Non-dyadic amounts: float32 level sums are order-sensitive -> checks the tick-ordered accumulation.
    tick=0.002 widens the bars beyond 128 / 512 levels (all LDS size classes)."""
    from finmlkit_amd.bar.base import comp_bar_footprints_csr
    ts, px, am, sd = orc.synth(13, 0, n)
    rng = np.random.default_rng(4)
    am = rng.lognormal(-1, 1.3, n).astype(dtype)
    sd = sd.copy()
    sd[rng.random(n) < 0.03] = 0
    _, ci = orc._time_bar_indexer(ts, interval)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], mult)
    off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], mult)
    _check_fp(off, flat, bar, woff, wflat, wbar, f"n={n} iv={interval}")


@pytest.mark.parametrize("case", ["t1", "t1_zeros", "t3_coarse", "t60", "sparse", "tail", "mult15"])
def test_footprints_lane_per_bar_schedule(orc, monkeypatch, case):
    """k_bar_footprints_lanes (one lane runs the reference's loops for one bar; forced here, the library picks it for >= 64 K
    bars of <= 64 ticks on average): non-dyadic float32 amounts (the float32 level sums are order-sensitive), 1-second bars
    with 1-8 levels, bars with more than 8 levels handed to the wave-per-bar kernel through the list (1-minute bars: all of
    them), unsigned ticks, empty bars, a stream that ends inside a 16-tick block."""
    from finmlkit_amd.bar.base import comp_bar_footprints_csr
    monkeypatch.setenv("FMK_FP_LANES", "2")
    n = 500_003 if case == "tail" else 400_000
    ts, px, am, sd = orc.synth(29, 0, n, 300_000_000_000) if case == "sparse" else orc.synth(29, 0, n)
    rng = np.random.default_rng(6)
    am = rng.lognormal(-1, 1.3, n).astype(np.float32)
    if case == "t1_zeros":
        sd = sd.copy()
        sd[rng.random(n) < 0.25] = 0
    interval = {"t60": 60.0, "t3_coarse": 3.0}.get(case, 1.0)
    tick = 0.05 if case == "t3_coarse" else 0.01
    _, ci = orc._time_bar_indexer(ts, interval)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    mult = 1.5 if case == "mult15" else 3.0
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], mult)
    off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], mult)
    _check_fp(off, flat, bar, woff, wflat, wbar, case)
    widths = np.diff(woff)
    print(f"{case}: {len(widths)} bars, levels per bar: mean {widths.mean():.2f}, max {widths.max()}, > 8: {(widths > 8).mean():.1%}")


def test_footprints_reference_shape_and_errors(orc):
    from finmlkit_amd.bar.base import comp_bar_footprints
    ts, px, am, sd = orc.synth(3, 0, 20_000)
    _, ci = orc._time_bar_indexer(ts, 60.0)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    got = comp_bar_footprints(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    want = orc.comp_bar_footprints(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    assert len(got) == 13 and len(got[0]) == len(ci) - 1
    for g, w in zip(got[:7], want[:7]):
        for a, b in zip(g, w):
            np.testing.assert_array_equal(a, b)
            assert a.dtype == b.dtype
    assert got[5][0].dtype == np.bool_
    # lows too high -> a tick falls outside the level range -> the reference's ValueError
    with pytest.raises(ValueError, match="Invalid price level index"):
        comp_bar_footprints(px, am, ci, sd, 0.01, o[2] + 0.05, o[1] + 0.05, 3.0)


@pytest.mark.parametrize("tick,interval,amounts", [(0.0001, 60.0, "dyadic"), (0.0001, 600.0, "lognormal"),
                                                  (0.00001, 3600.0, "dyadic")])
def test_footprints_wide_bars_global_histogram(orc, tick, interval, amounts):
    """Bars wider than 2048 levels (a fine tick on a coarse price grid): histogram in global scratch, same results."""
    from finmlkit_amd.bar.base import comp_bar_footprints_csr
    n = 150_000
    ts, px, am, sd = orc.synth(31, 0, n)
    if amounts == "lognormal":
        am = np.random.default_rng(4).lognormal(-1, 1.0, n).astype(np.float32)
    _, ci = orc._time_bar_indexer(ts, interval)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], 3.0)
    assert np.diff(woff).max() > 2048
    off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], 3.0)
    _check_fp(off, flat, bar, woff, wflat, wbar, f"wide tick={tick}")


def test_imbalance_flags_compare_in_float64(orc):
    """Decimal lots: sell 0.3, buy 0.1 (float32 level sums), factor 3.0.  The reference's production path is Numba-typed:
    array(float32) * float64 is float64, so 0.30000001192 > 0.1f * 3.0 = 0.30000000447 is True; NumPy 2 (NEP 50, the
    pure-Python mode) rounds the product to float32 and gets 0.3f > 0.3f = False.  Oracle and kernel follow the typed
    semantics (every recorded reference call has exact products, where the two agree)."""
    from finmlkit_amd.bar.base import comp_footprint_features
    lv = np.arange(100, 104, dtype=np.int32)
    buy = np.array([0.5, 0.1, 0.9, 0.1], dtype=np.float32)
    sell = np.array([0.3, 0.2, 0.3, 0.7], dtype=np.float32)
    assert not (np.float32(0.3) > np.float32(0.1) * np.float32(3.0)) and float(np.float32(0.3)) > float(np.float32(0.1)) * 3.0
    got = comp_footprint_features(lv, buy, sell, 3.0)
    want = orc.comp_footprint_features(lv, buy, sell, 3.0)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    assert list(got[1]) == [True, False, True, False]                   # sell[l] > buy[l+1] * 3: levels 0 and 2 are the 0.3 vs 0.1 pairs
    assert got[2] == want[2] and got[3] == want[3]


@pytest.mark.parametrize("amounts", ["dyadic", "lognormal32", "f64", "ties", "signflip", "huge_then_small"])
@pytest.mark.parametrize("rows", ["1", "2"])
def test_order_flow_redo_in_tick_order(orc, monkeypatch, amounts, rows):
    """Every bar forced onto the redo list (FMK_DIR_FORCE_REDO=1) and redone in tick order: by the row-per-wave kernel, which adds
    64 terms per step with the machine's own rounding to the running sum's grid (exact unless a term ties or the sum leaves its
    binade -- then term by term), and by the wave-per-bar lane walkers (FMK_DIR_FORCE_REDO=2).  Both must give the oracle's sequential sums
    bit for bit: short bars (the sum climbs through many binades), long bars, signed running sums that cross zero, exact ties
    (terms that are half a grid step: dyadic prices), a huge first term followed by tiny ones, float64 amounts."""
    from finmlkit_amd import engine
    monkeypatch.setenv("FMK_DIR_FORCE_REDO", rows)
    monkeypatch.setenv("FMK_DIR_LANES", "0")
    n = 400_000
    ts, px, am, sd = orc.synth(31, 0, n)
    rng = np.random.default_rng(17)
    if amounts == "lognormal32":
        am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    elif amounts == "f64":
        am = rng.lognormal(-1, 1.2, n)
    elif amounts == "ties":
        px = np.round(px * 4) / 4 + 64.0                                  # dyadic prices: products with few bits -> exact half steps
        am = (rng.integers(1, 1 << 20, n) * 2.0 ** -12).astype(np.float64)
    elif amounts == "signflip":
        sd = np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.int8)       # the signed sums hover around zero
        am = rng.lognormal(-1, 0.2, n).astype(np.float32)
    elif amounts == "huge_then_small":
        am = rng.lognormal(-8, 1.0, n).astype(np.float64)
        am[::50_000] = 1e9
    ci = np.concatenate([[-1], np.sort(rng.choice(n - 1, 60, replace=False)), [n - 1]]).astype(np.int64)   # 61 bars, 1 .. 30 000 ticks
    t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
    d, nz = t.bar_directional(engine.DeviceArray.from_host(t.ctx, ci))
    want = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    got = engine.to_host(d)
    for k, w in zip(G.DIR_KEYS, want):
        np.testing.assert_array_equal(got[k], w, err_msg=f"{k} ({amounts}, rows={rows})")


def test_order_flow_on_negative_prices(orc):
    """A tape whose prices are NEGATIVE (spreads, some futures): the dollar sums are negative, and the tie test of the wave-per-bar
    kernel bounded the two orders' difference by eps * sum -- a negative bound: no bar was ever redone and ~0.3 % of the bars kept the
    parallel order's last float32 bit (tools/fuzz_fused.py seed 7707, case 9).  The bound takes magnitudes now."""
    from finmlkit_amd.bar.base import comp_bar_directional_features
    rng = np.random.default_rng(7707)
    n = 3_000_000
    px = np.round(-40.0 + np.cumsum(rng.integers(-1, 2, n)) * 0.0005, 4)
    assert px.max() < 0
    am = (rng.integers(1, 4097, n) / 1024.0).astype(np.float32)
    sd = rng.choice(np.array([-1, 1], np.int8), n)
    lens = np.maximum(1, rng.normal(900, 45, int(n / 900 * 1.2)).astype(np.int64))
    ci = np.concatenate([[-1], np.cumsum(lens) - 1])
    ci = ci[ci <= n - 1].astype(np.int64)
    got = comp_bar_directional_features(px, am, ci, sd)
    want = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    _check_dir(got, want, "negative prices")


@pytest.mark.parametrize("f64", [False, True])
def test_order_flow_on_prices_that_change_sign_inside_a_bar(orc, f64):
    """Prices (and, with float64 amounts, sizes) of BOTH signs inside one bar: the dollar sums cancel, so |dollars_buy| + |dollars_sell|
    is no longer the sum of the terms' magnitudes the tie bounds are built on.  The kernel carries that sum itself and sends
    every bar in which it is larger to the tick-order redo; bars of ~900 ticks (one wave), a 40 000-tick bar (eight waves, composed) and
    the few-tick bars between them, against the oracle on all 14 columns."""
    from finmlkit_amd.bar.base import comp_bar_directional_features
    rng = np.random.default_rng(7711)
    n = 2_000_000
    px = np.round(np.cumsum(rng.integers(-1, 2, n)) * 0.0005, 4) + rng.choice([-0.25, 0.25], n)     # hovers around zero, both signs in every bar
    px[px == 0.0] = 0.0005
    if f64:
        am = rng.integers(1, 4097, n) / 1024.0 * rng.choice([1.0, 1.0, 1.0, -1.0], n)               # a quarter of the sizes negative
        am = am + rng.random(n) * 2.0 ** -30                                                        # full float64 mantissas
    else:
        am = (rng.integers(1, 4097, n) / 1024.0).astype(np.float32)
    sd = rng.choice(np.array([-1, 1], np.int8), n)
    lens = np.maximum(1, rng.normal(900, 45, int(n / 900 * 1.2)).astype(np.int64))
    lens[5] = 40_000
    lens[9:12] = (1, 2, 3)
    ci = np.concatenate([[-1], np.cumsum(lens) - 1])
    ci = ci[ci <= n - 1].astype(np.int64)
    assert (px[: ci[8]] > 0).any() and (px[: ci[8]] < 0).any()
    got = comp_bar_directional_features(px, am, ci, sd)
    want = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    _check_dir(got, want, f"prices of both signs inside the bars, float64 amounts {f64}")


def test_order_flow_extremum_near_tie_with_nan_elsewhere_in_the_bar(orc):
    """tools/fuzz_longbars.py seed 361, case 70 (tests/golden/nan_tie_longbar.npz, tools/gen_nan_tie_fixture.py): a 65 537-tick bar whose running
    signed dollar sum peaks 2.7e-12 below a float32 rounding boundary at tick 488 and holds a NaN amount at tick 554.  The NaN made the
    error bound of the tie test NaN (it contains the bar's dollar total), `distance <= NaN` said "not near", and the bar kept the
    parallel order's last bit: cum_dollars_max 6534.9214 against the reference's 6534.921.  An undefined bound now means redo."""
    from finmlkit_amd.bar.base import comp_bar_directional_features
    px, am, ci, sd, want = G.nan_tie_longbar()
    got = comp_bar_directional_features(px, am, ci, sd)
    _check_dir(got, want, "nan_tie_longbar")


@pytest.mark.parametrize("amounts", ["dyadic", "dyadic_heavy", "lognormal32", "f64_dyadic", "f64", "negative", "nan", "zeros",
                                     "quantum_changes"])
def test_footprints_long_bars_workgroup_per_bar(orc, amounts):
    """Bars of more than 8 192 ticks (k_bar_footprints_wide: sixteen waves on one LDS histogram, integer units, certified per
    (level, side) key) next to short ones in one call: dyadic amounts whose BAR total is far beyond 2^24 units while every key
    stays below it (exact, order-free); amounts heavy enough that a key passes 2^24 units (the float32 sums round: tick order);
    full-mantissa float32 and float64 amounts (tick order); a negative and a NaN amount inside a long bar; all-zero amounts; a
    stream whose quantum changes from bar to bar (the statistics pass); unsigned ticks; a bar of more than 2 048 levels among
    the long ones (stays with the wave kernel's global-scratch class)."""
    from finmlkit_amd.bar.base import comp_bar_footprints_csr
    n = 700_000
    ts, px, am, sd = orc.synth(37, 0, n)
    rng = np.random.default_rng(12)
    sd = sd.copy()
    sd[rng.random(n) < 0.02] = 0
    if amounts == "dyadic_heavy":
        am = (rng.integers(1, 1 << 16, n) * 2.0 ** -4).astype(np.float32)
    elif amounts == "lognormal32":
        am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    elif amounts == "f64_dyadic":
        am = (rng.integers(1, 4096, n) * 2.0 ** -10).astype(np.float64)
    elif amounts == "f64":
        am = rng.lognormal(-1, 1.2, n)
    elif amounts == "negative":
        am = am.copy(); am[123_456] = -0.5
    elif amounts == "nan":
        am = am.copy(); am[223_456] = np.nan
    elif amounts == "zeros":
        am = np.zeros(n, np.float32)
    elif amounts == "quantum_changes":
        am = am.copy()
        am[100_000:300_000] = (rng.integers(1, 64, 200_000) * 2.0 ** -3).astype(np.float32)
        am[300_000:420_000] = (rng.integers(1, 1 << 12, 120_000) * 2.0 ** -14).astype(np.float32)
    # bars: 9 000 .. 160 000 ticks (16 384 and 16 385 among them: the last wave-kernel length and the first workgroup one), a few
    # short ones and an empty one in between
    cuts = [-1, 50, 9_100, 9_100, 25_484, 25_484 + 16_385, 61_000, 210_000, 225_000, 420_000, 430_000, 520_000, 690_000, n - 1]
    ci = np.array(cuts, dtype=np.int64)
    tick = 0.01
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], 3.0)
    off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], 3.0)
    _check_fp(off, flat, bar, woff, wflat, wbar, amounts)
    if amounts in ("dyadic", "lognormal32", "f64"):
        # finer ticks: the widest bar decides how many tick segments the tick-ordered path can scatter concurrently (16 counter arrays
        # fit the LDS up to ~600 levels, 1 beyond ~5 000) and beyond 6 144 levels a long bar stays with the wave kernel's
        # global-scratch class
        for fine in (0.002, 0.0005, 0.0002):
            woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, fine, o[2], o[1], 3.0)
            off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, fine, o[2], o[1], 3.0)
            _check_fp(off, flat, bar, woff, wflat, wbar, f"{amounts}, tick {fine}: widest bar {np.diff(woff).max()} levels")
        assert np.diff(woff).max() > 6144


@pytest.mark.parametrize("long_bars", [False, True])
def test_cot_with_nan_level_sums(orc, long_bars):
    """np.argmax takes the FIRST NaN as the maximum (base.py:829): a NaN size on a middle level moves the centre of trades there --
    wave-per-bar and workgroup-per-bar kernels, and the lane-per-bar kernel on short bars."""
    from finmlkit_amd.bar.base import comp_bar_footprints_csr
    rng = np.random.default_rng(3)
    n = 60_000 if long_bars else 6000
    px = 100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, n))
    sd = rng.choice(np.array([-1, 1], np.int8), n)
    am = rng.lognormal(-1, 1, n).astype(np.float32)
    am[[n // 3, n // 2 + 5]] = np.nan
    if long_bars:
        ci = np.array([-1, 20_000, 45_000, n - 1], np.int64)
    else:
        ci = np.concatenate([[-1], np.arange(19, n, 20)]).astype(np.int64)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    _check_fp(off, flat, bar, woff, wflat, wbar, f"nan levels, long={long_bars}")
    b = int(np.searchsorted(ci, n // 3, side="left")) - 1
    tv = wflat["buy_volumes"][woff[b]:woff[b + 1]] + wflat["sell_volumes"][woff[b]:woff[b + 1]]
    assert np.isnan(tv).any() and wbar["cot_price_levels"][b] == wflat["price_levels"][woff[b]:woff[b + 1]][np.argmax(tv)]


@pytest.mark.parametrize("w,amounts", [(8, "dyadic"), (25, "dyadic"), (25, "full"), (70, "dyadic")])
def test_footprints_bars_of_many_levels(orc, w, amounts):
    """Bars of ~1 200 ticks whose price path covers hundreds to thousands of levels (steps of up to w ticks: a fine price_tick_size on a
    fast market): the level classes beyond 256 -- 512, the two-wave 1 024 class, 2 048, the global-scratch class -- with the two
    np.sum over the levels by the parallel tree routine (fp_emit_bar's fast_sum); dyadic sizes (integer-unit path) and full-mantissa
    ones (tick order).  Every array against the oracle."""
    from finmlkit_amd.bar.base import comp_bar_footprints_csr, comp_bar_ohlcv
    rng = np.random.default_rng(100 + w)
    n = 90_000
    px = np.round(np.maximum(60000.0 + 0.01 * np.cumsum(rng.integers(-w, w + 1, n)), 1.0), 2)
    am = (rng.integers(1, 4097, n) * 2.0 ** -10).astype(np.float32) if amounts == "dyadic" else rng.lognormal(-1, 1.2, n).astype(np.float32)
    sd = rng.choice(np.array([-1, 1], dtype=np.int8), n)
    lens = [int(v) for v in rng.integers(600, 2400, 60)]
    ci = np.cumsum([-1] + lens).astype(np.int64)
    ci = ci[ci <= n - 1]
    o = orc.comp_bar_ohlcv(px, am, ci)
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    lev = np.diff(woff)
    off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    _check_fp(off, flat, bar, woff, wflat, wbar, f"w={w} {amounts}: levels per bar {int(lev.min())}..{int(lev.max())}")
    if w == 25 and amounts == "full":
        # ... and through the fused cfg-4 call (same fill, its own size pass)
        from finmlkit_amd import engine
        ts = 1_700_000_000_000_000_000 + np.arange(n, dtype=np.int64) * 50_000_000
        t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
        o2, dd, nz, off2, flat2, bar2, bad = t.bars_fused(engine.DeviceArray.from_host(t.ctx, ci), 0.01, 3.0, want_median=True)
        assert int(bad.to_host()[0]) == 0
        _check_fp(off2.to_host(), engine.to_host(flat2), engine.to_host(bar2), woff, wflat, wbar, "fused, many levels")
    if w == 8:
        assert lev.max() > 256 and lev.min() < 512
    if w == 25:
        assert (lev > 1024).any() and (lev <= 1024).any() and (lev > 512).any()
    if w == 70:
        assert (lev > 2048).any()


@pytest.mark.parametrize("amounts", ["dyadic", "full"])
def test_footprints_exact_level_class_edges(orc, amounts):
    """Bars whose level count sits exactly on and one beyond every class edge of the wave kernel (128, 256, 512, 768, 1 024, 1 536,
    2 048, 3 072, 4 096: the 24 B / 16 B per level layouts, two and one wave per workgroup, LDS and global scratch), bars of 300 .. 5 000
    ticks in random order.  Every array against the oracle."""
    from finmlkit_amd.bar.base import comp_bar_footprints_csr
    rng = np.random.default_rng(77)
    edges = [127, 128, 129, 255, 256, 257, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, 1535, 1536, 1537, 2047, 2048, 2049, 3071,
             3072, 3073, 4095, 4096, 4097, 5000, 1, 2]
    rng.shuffle(edges)
    px_parts, lens = [], []
    for L in edges:
        m = int(rng.integers(max(300, L // 2), 5000)) if L > 2 else int(rng.integers(1, 50))
        lv = rng.integers(0, L, m)
        lv[rng.integers(0, m)] = 0
        if m > 1:
            j = int(rng.integers(0, m))
            lv[j if lv[j] != 0 or L == 1 else (j + 1) % m] = L - 1
            if not (lv == 0).any(): lv[0] = 0
            if not (lv == L - 1).any(): lv[-1] = L - 1
        else:
            lv[0] = 0
        px_parts.append(1000.0 + 0.01 * lv)
        lens.append(m)
    px = np.round(np.concatenate(px_parts), 2)
    n = len(px)
    am = (rng.integers(1, 4097, n) * 2.0 ** -10).astype(np.float32) if amounts == "dyadic" else rng.lognormal(-1, 1.2, n).astype(np.float32)
    sd = rng.choice(np.array([-1, 1], dtype=np.int8), n)
    ci = np.cumsum([-1] + lens).astype(np.int64)
    o = orc.comp_bar_ohlcv(px, am, ci)
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    lev = np.diff(woff)
    assert set(int(v) for v in lev) >= {128, 129, 512, 513, 768, 769, 1024, 1025, 1536, 1537, 2048, 2049, 3072, 3073, 4096, 4097}, sorted(lev)
    off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    _check_fp(off, flat, bar, woff, wflat, wbar, f"exact level edges, {amounts}")
