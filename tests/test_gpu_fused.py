"""GPU parity of the cfg-4 fused path (fmk_bars_flow_size_defer_dev + the footprint fill): OHLCV, then order-flow +
footprints from ONE read of price/amount/side by the two-waves-per-bar kernel.  Checked against the CPU oracle, the
reference-generated goldens and the separate reducers (same arithmetic -> identical bits)."""
import numpy as np
import pytest

from tests import _golden as G
from tests.test_gpu_features import INT_DIR, _check_dir, _check_fp

pytestmark = pytest.mark.gpu

OHLCV_KEYS = ["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"]


def _fused(px, am, sd, ci, tick=0.01, imb=3.0):
    from finmlkit_amd import engine
    t = engine.DeviceTrades.from_numpy(np.zeros(len(px), np.int64), px, am, sd)
    cid = engine.DeviceArray.from_host(t.ctx, np.ascontiguousarray(ci, dtype=np.int64))
    o, d, nz, off, flat, bar, bad = t.bars_fused(cid, tick, imb)
    return (t, cid, engine.to_host(o), engine.to_host(d), int(nz.to_host()[0]), off.to_host(), engine.to_host(flat),
            engine.to_host(bar), int(bad.to_host()[0]))


def _check_all(orc, px, am, sd, ci, what, tick=0.01):
    t, cid, o, d, nz, off, flat, bar, bad = _fused(px, am, sd, ci, tick)
    assert bad == 0
    want_o = orc.comp_bar_ohlcv(px, am, ci)
    for k, w in zip(OHLCV_KEYS, want_o):
        if k == "vwap":
            G.assert_f64_close(o[k], w, rtol=1e-9, what=f"{what}:vwap")
        else:
            np.testing.assert_array_equal(o[k], w, err_msg=f"{what}:{k}")
    want_d = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    assert nz == int(np.isnan(want_d[6]).sum())
    _check_dir(tuple(d[k] for k in G.DIR_KEYS), want_d, what)
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, tick, want_o[2], want_o[1], 3.0)
    _check_fp(off, flat, bar, woff, wflat, wbar, what)
    # the separate reducers run the same arithmetic: bit-identical
    from finmlkit_amd import engine
    d2, _ = t.bar_directional(cid)
    for k, v in engine.to_host(d2).items():
        np.testing.assert_array_equal(d[k], v, err_msg=f"{what}: fused vs separate {k}")
    lows = engine.DeviceArray.from_host(t.ctx, o["low"])
    highs = engine.DeviceArray.from_host(t.ctx, o["high"])
    off2, flat2, bar2, _ = t.bar_footprints(cid, lows, highs, tick, 3.0)
    np.testing.assert_array_equal(off, off2.to_host())
    for k, v in {**engine.to_host(flat2), **engine.to_host(bar2)}.items():
        got = flat[k] if k in flat else bar[k]
        np.testing.assert_array_equal(got, v, err_msg=f"{what}: fused vs separate {k}")


@pytest.mark.parametrize("case", ["syn_t60", "syn_t1", "syn_tick100", "syn_vol2048", "rnd_t120", "rnd_tick37"])
def test_fused_golden(orc, case):
    d = G.load("reducers")
    px, am, sd = G.reducer_stream(orc, d, case)
    ci = d[f"{case}__ci"]
    t, cid, o, dr, nz, off, flat, bar, bad = _fused(px, am, sd, ci)
    assert bad == 0 and nz == 0
    _check_dir(tuple(dr[k] for k in G.DIR_KEYS), tuple(d[f"{case}__dir_{k}"] for k in G.DIR_KEYS), case)
    _check_fp(off, flat, bar, d[f"{case}__fp_offsets"], {k: d[f"{case}__fp_{k}"] for k in G.FP_LIST_KEYS},
              {k: d[f"{case}__fp_{k}"] for k in G.FP_BAR_KEYS}, case)
    np.testing.assert_array_equal(o["low"], d[f"{case}__ohlcv_low"])


@pytest.mark.parametrize("n,interval,amounts,zeros", [
    (400_000, 60.0, "dyadic", False),        # exact integer-unit footprint path
    (300_000, 60.0, "lognormal32", False),   # float32 amounts with inexact sums: tick-ordered path
    (200_000, 60.0, "f64", True),            # float64 amounts, unsigned ticks
    (300_000, 1.0, "dyadic", True),          # tiny bars (1, 2, 4 ticks per lane tiles), unsigned ticks
    (300_000, 7200.0, "dyadic", False),      # long bars: many tiles, > 128 levels -> streaming footprint kernel
    (150_000, 7200.0, "lognormal32", False),
    (100_000, 3.0, "mixed", False),          # dyadic with a few inexact amounts: exact -> retry / ordered switches
    (130, 60.0, "dyadic", False), (1, 60.0, "dyadic", False),
])
@pytest.mark.parametrize("flow_lanes", ["1", "2"])
def test_fused_vs_oracle(orc, monkeypatch, n, interval, amounts, zeros, flow_lanes):
    """flow_lanes 2: the first half of cfg 4 through k_bar_dir_lanes<OHLC> + the median-only small-bar kernel whatever the number
    of bars (the library takes that path for >= 65 536 bars of 600..2048 ticks on average; bars beyond 8 192 ticks go to the
    wave-per-bar kernels by list / flag)."""
    monkeypatch.setenv("FMK_FLOW_LANES", flow_lanes)
    ts, px, am, sd = orc.synth(17, 0, n)
    rng = np.random.default_rng(3)
    if amounts == "lognormal32":
        am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    elif amounts == "f64":
        am = rng.lognormal(-1, 1.2, n)
    elif amounts == "mixed":
        am = am.copy()
        am[rng.random(n) < 0.002] = np.float32(0.3)
    if zeros:
        sd = sd.copy()
        sd[rng.random(n) < 0.1] = 0
    if n > 1:
        _, ci = orc._time_bar_indexer(ts, interval)
    else:
        ci = np.array([-1, 0], dtype=np.int64)
    _check_all(orc, px, am, sd, ci, f"n={n} iv={interval} {amounts}")


@pytest.mark.parametrize("flow_lanes", ["1", "2"])
def test_fused_sparse_stream_empty_bars(orc, monkeypatch, flow_lanes):
    monkeypatch.setenv("FMK_FLOW_LANES", flow_lanes)
    ts, px, am, sd = orc.synth(42, 0, 20_000, orc.SPARSE_GAP_MOD)
    _, ci = orc._time_bar_indexer(ts, 60.0)
    assert (np.diff(ci) == 0).any()
    _check_all(orc, px, am, sd, ci, "sparse")


def test_fused_bad_level_and_errors(orc):
    from finmlkit_amd import _ffi
    ts, px, am, sd = orc.synth(2, 0, 50_000)
    _, ci = orc._time_bar_indexer(ts, 60.0)
    with pytest.raises(ValueError):
        _fused(px, am, sd, ci[:1])
    with pytest.raises(ValueError):
        _fused(px, am, sd, ci, tick=0.0)
    # a coarser tick than the data's: levels stay inside [low, high], nothing is flagged
    t, cid, o, d, nz, off, flat, bar, bad = _fused(px, am, sd, ci, tick=0.05)
    assert bad == 0
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.05, o["low"], o["high"], 3.0)
    _check_fp(off, flat, bar, woff, wflat, wbar, "tick 0.05")


@pytest.mark.parametrize("n,interval,amounts", [
    (400_000, 60.0, "dyadic"), (300_000, 60.0, "lognormal32"),
    (300_000, 60.0, "ties"),        # a size most trades share: the degenerate bracket (blo == bhi, no candidate list)
    (300_000, 60.0, "two_values"),  # the two middle ranks straddle two heavy ties
    (300_000, 60.0, "drift"),       # the size distribution doubles every few bars: bracket misses -> generic selection
    (200_000, 60.0, "nan"),         # NaN sizes: np.median is NaN for those bars
    (300_000, 150.0, "dyadic"),     # 3 000-tick bars: beyond the in-sweep limit, flagged for the long-bar median kernels
    (60_000, 13.0, "lognormal32"),  # 260-tick bars
])
def test_fused_median_taken_by_the_footprint_sweep(orc, monkeypatch, n, interval, amounts):
    """FMK_FLOW_MEDIAN_DEFER=1 (off by default, profiles/r03_cfg4.txt): pass 1 leaves the median trade size to the footprint sweep
    (fmk_bars_flow_size_defer_dev -> fmk_comp_bar_footprints_fill_median_dev), which brackets the middle ranks from the wave's
    previous bar and selects them exactly among the candidates -- np.median's bits whatever the bracket does."""
    monkeypatch.setenv("FMK_FLOW_LANES", "2")
    monkeypatch.setenv("FMK_FLOW_MEDIAN_DEFER", "1:blocks=2")        # 2 workgroups = 8 waves: every wave carries its bracket over ~30 bars
    ts, px, am, sd = orc.synth(23, 0, n)
    rng = np.random.default_rng(5)
    if amounts == "lognormal32":
        am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    elif amounts == "ties":
        am = np.where(rng.random(n) < 0.7, np.float32(0.001), rng.lognormal(-3, 2.0, n)).astype(np.float32)
    elif amounts == "two_values":
        am = np.where(rng.random(n) < 0.5, np.float32(0.25), np.float32(0.5)).astype(np.float32)
    elif amounts == "drift":
        am = (rng.lognormal(-1, 0.3, n) * 2.0 ** (np.arange(n) // 9_000 % 7)).astype(np.float32)
    elif amounts == "nan":
        am = rng.lognormal(-1, 1.2, n).astype(np.float32)
        am[rng.integers(0, n, 40)] = np.nan
    _, ci = orc._time_bar_indexer(ts, interval)
    from finmlkit_amd import _ffi
    from finmlkit_amd._ffi import c_i64
    import ctypes as C
    t, cid, o, d, nz, off, flat, bar, bad = _fused(px, am, sd, ci)
    want = orc.comp_bar_ohlcv(px, am, ci)
    np.testing.assert_array_equal(o["median_trade_size"], want[7])          # NaN positions included
    fb = c_i64()
    t.ctx.call("fmk_diag_fp_median_fallbacks", C.byref(fb))
    print(f"{amounts} @ {interval:g} s: {fb.value} of {len(ci) - 1} bars took the generic selection")
    if amounts in ("dyadic", "lognormal32", "ties") and interval == 60.0:
        assert fb.value <= 8 + (len(ci) - 1) // 20              # the bracket is accepted after each wave's first bar
    if amounts != "nan":
        _check_all(orc, px, am, sd, ci, f"deferred median {amounts}")


@pytest.mark.parametrize("sort", ["0", "1", "census"])
@pytest.mark.parametrize("amounts", ["dyadic", "lognormal32"])
def test_fused_on_lognormal_bar_lengths_sorted_lanes(orc, monkeypatch, sort, amounts):
    """cfg 4 on bars of UNEQUAL length (lognormal, sigma 1, mean ~900 ticks: what real one-minute bars look like), with and without the
    length-ordered lane schedule (FMK_FLOW_SORT=1: a counting sort of the bars by quarter-octave length class, longest first, feeds
    k_bar_dir_lanes) -- every output against the oracle; FMK_FLOW_LANES=2 forces the lane-per-bar first half at this size."""
    if sort == "census":
        monkeypatch.delenv("FMK_FLOW_SORT", raising=False)       # the library decides from its census of the bar lengths (here: sort)
    else:
        monkeypatch.setenv("FMK_FLOW_SORT", sort)
    monkeypatch.setenv("FMK_FLOW_LANES", "2")
    rng = np.random.default_rng(77)
    n = 1_500_000
    lens = np.maximum(1, rng.lognormal(np.log(900.0) - 0.5, 1.0, int(n / 900 * 1.4)).astype(np.int64))
    lens[rng.integers(0, len(lens), 12)] = 0                              # a few empty bars
    lens[rng.integers(0, len(lens), 3)] = rng.integers(8193, 20000, 3)    # and some beyond the lane schedule's reach
    ci = np.concatenate([[-1], np.cumsum(lens) - 1])
    ci = ci[ci <= n - 1].astype(np.int64)
    px = np.round(100.0 + np.cumsum(rng.integers(-1, 2, n)) * 0.01, 2)
    sd = rng.choice(np.array([-1, 1], np.int8), n)
    am = (rng.integers(1, 4097, n) / 1024.0).astype(np.float32) if amounts == "dyadic" else rng.lognormal(-1, 1.2, n).astype(np.float32)
    _check_all(orc, px, am, sd, ci, f"lognormal lengths, sort {sort}, {amounts}")


@pytest.mark.parametrize("n,interval,amounts,zeros", [
    (400_000, 60.0, "dyadic", False),        # ~500-tick bars, integer units certify
    (300_000, 60.0, "lognormal32", False),   # sizes that do not certify: float64 volumes
    (300_000, 60.0, "mixed", True),          # a few inexact sizes among dyadic ones, unsigned ticks
    (300_000, 7.0, "dyadic", False),         # bars of a few dozen ticks: one tick per lane and fewer
    (400_000, 900.0, "dyadic", False),       # bars of several tiles: k_fu_long
    (130, 60.0, "dyadic", False),
])
@pytest.mark.parametrize("fused", ["0", "2", "3"])
def test_one_pass_kernels_forced(orc, monkeypatch, n, interval, amounts, zeros, fused):
    """FMK_FUSED: cfg 4 through the one-pass kernels of csrc/fmk_fused.h whatever the tape looks like -- 2: with the integer-unit
    certificate (a tape that does not certify falls through to the two-pass form by itself), 3: with float64 volumes -- or never (0).
    The library picks by bar count, mean bar length and a census of the bar lengths (bars_flow_fused_ok), which the small tapes of this
    suite never pass: forced, every case must give the oracle's bits, and the separate reducers' bits (_check_all)."""
    monkeypatch.setenv("FMK_FUSED", fused)
    ts, px, am, sd = orc.synth(29, 0, n)
    rng = np.random.default_rng(11)
    if amounts == "lognormal32":
        am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    elif amounts == "mixed":
        am = am.copy()
        am[rng.random(n) < 0.002] = np.float32(0.3)
    if zeros:
        sd = sd.copy()
        sd[rng.random(n) < 0.1] = 0
    _, ci = orc._time_bar_indexer(ts, interval)
    _check_all(orc, px, am, sd, ci, f"FMK_FUSED={fused} n={n} iv={interval} {amounts}")
