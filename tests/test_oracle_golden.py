"""Pin the CPU oracle (oracle/fmk_oracle.c) against fixtures produced by the reference itself.

CPU-only.  Integer outputs must be bit-exact; float64 outputs are sequential restatements of
the same arithmetic, so they are checked for exact equality too except where NumPy evaluates
through BLAS/libm paths we cannot reproduce bit-for-bit (stated per assert).
"""
import numpy as np
import pytest

from tests import _golden as G


def test_time_indexer_golden(orc):
    d = G.load("time_indexer")
    for c in G.cases(d):
        ts = d[f"{c}__ts"] if f"{c}__ts" in d else G.synth_from(orc, d[f"{c}__synth"])[0]
        clock, idx = orc._time_bar_indexer(ts, float(d[f"{c}__interval"]))
        np.testing.assert_array_equal(clock, d[f"{c}__clock"], err_msg=c)
        np.testing.assert_array_equal(idx, d[f"{c}__idx"], err_msg=c)


def test_threshold_indexers_golden(orc):
    d = G.load("threshold_indexers")
    ts, px, am, sd = G.synth_from(orc, d["synth"])
    for k, want in d.items():
        kind, _, thr = k.partition("_")
        if kind == "tick":
            got = orc._tick_bar_indexer(ts, int(thr))
        elif kind == "vol32":
            got = orc._volume_bar_indexer(am, float(thr))
        elif kind == "dol32":
            got = orc._dollar_bar_indexer(px, am, float(thr))
        elif kind == "vol64":
            got = orc._volume_bar_indexer(d["r_am"], float(thr))
        elif kind == "dol64":
            got = orc._dollar_bar_indexer(d["r_px"], d["r_am"], float(thr))
        else:
            continue
        np.testing.assert_array_equal(got, want, err_msg=k)
    np.testing.assert_array_equal(orc._dollar_bar_indexer(d["big_px"], d["big_am"], 100.0), d["big_dol_100"])
    np.testing.assert_array_equal(orc._volume_bar_indexer(d["big_am"], 10.0), d["big_vol_10"])


REDUCER_CASES = ["syn_t60", "syn_t1", "syn_tick100", "syn_vol2048", "rnd_t120", "rnd_tick37", "sparse_t60"]


@pytest.mark.parametrize("case", REDUCER_CASES)
def test_ohlcv_golden(orc, case):
    d = G.load("reducers")
    px, am, sd = G.reducer_stream(orc, d, case)
    got = orc.comp_bar_ohlcv(px, am, d[f"{case}__ci"])
    for k, g in zip(G.OHLCV_KEYS, got):
        np.testing.assert_array_equal(g, d[f"{case}__ohlcv_{k}"], err_msg=f"{case}:{k}")
        assert g.dtype == d[f"{case}__ohlcv_{k}"].dtype, k


@pytest.mark.parametrize("case", [c for c in REDUCER_CASES if c != "sparse_t60"])
def test_directional_golden(orc, case):
    d = G.load("reducers")
    px, am, sd = G.reducer_stream(orc, d, case)
    got = orc.comp_bar_directional_features(px, am, d[f"{case}__ci"], sd)
    for k, g in zip(G.DIR_KEYS, got):
        np.testing.assert_array_equal(g, d[f"{case}__dir_{k}"], err_msg=f"{case}:{k}")
        assert g.dtype == d[f"{case}__dir_{k}"].dtype, k


@pytest.mark.parametrize("case", [c for c in REDUCER_CASES if c != "sparse_t60"])
def test_footprints_golden(orc, case):
    d = G.load("reducers")
    px, am, sd = G.reducer_stream(orc, d, case)
    ci = d[f"{case}__ci"]
    off, flat, bar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, d[f"{case}__ohlcv_low"],
                                                 d[f"{case}__ohlcv_high"], 3.0)
    np.testing.assert_array_equal(off, d[f"{case}__fp_offsets"])
    for k in G.FP_LIST_KEYS:
        np.testing.assert_array_equal(flat[k].astype(d[f"{case}__fp_{k}"].dtype), d[f"{case}__fp_{k}"],
                                      err_msg=f"{case}:{k}")
    for k in G.FP_BAR_KEYS:
        want = d[f"{case}__fp_{k}"]
        if k == "vp_skew":
            # sum((p - vwap) * v) / sum(v) is identically 0 in exact arithmetic: the reference value
            # is rounding noise of a BLAS dot product -> absolute tolerance scaled by the level index.
            np.testing.assert_allclose(bar[k], want, rtol=0, atol=1e-6, err_msg=f"{case}:{k}")
        else:
            np.testing.assert_array_equal(bar[k], want, err_msg=f"{case}:{k}")


def test_footprint_features_golden(orc):
    d = G.load("footprint_features")
    for c in G.cases(d):
        bi, si, run, cot, sk, gi = orc.comp_footprint_features(d[f"{c}__lv"], d[f"{c}__b"], d[f"{c}__s"], 1.5)
        np.testing.assert_array_equal(bi, d[f"{c}__bi"], err_msg=c)
        np.testing.assert_array_equal(si, d[f"{c}__si"], err_msg=c)
        wrun, wcot, wsk, wgi = d[f"{c}__scalars"]
        assert run == int(wrun) and cot == int(wcot), c
        assert gi == wgi, (c, gi, wgi)                 # float32 pairwise order pinned bit-exactly
        assert abs(sk - wsk) <= 1e-9 * 2000, (c, sk, wsk)   # rounding noise, see above


def test_ticklevel_golden(orc):
    d = G.load("ticklevel")
    ts, px, am, sd = G.synth_from(orc, d["synth"])
    for w in (1e-6, 0.5, 5.0, 60.0):
        for lg in (0, 1):
            got = orc.comp_lagged_returns(ts, px, w, bool(lg))
            want = d[f"ret_{w}_{lg}"]
            if lg:   # libm log vs NumPy's SIMD log: <=1 ulp apart
                G.assert_f64_close(got, want, rtol=1e-12, what=f"ret {w} log")
                assert np.array_equal(np.isnan(got), np.isnan(want))
            else:
                np.testing.assert_array_equal(got, want, err_msg=f"ret {w}")
    np.testing.assert_array_equal(orc.comp_lagged_returns(d["small_ts"], d["small_px"], 2.0, False),
                                  d["small_ret_2.0_0"])
    r = orc.comp_lagged_returns(ts, px, 5.0, True)
    r = d["ret_5.0_1"]   # use the reference's returns so exp/log ulp noise does not stack
    for hl in (1.0, 30.0, 600.0):
        G.assert_f64_close(orc.ewmst(ts, r, hl), d[f"ewmst_{hl}"], rtol=1e-11, what=f"ewmst {hl}")
        G.assert_f64_close(orc.ewmst_mean0(ts, r, hl), d[f"ewmst0_{hl}"], rtol=1e-11, what=f"ewmst0 {hl}")
    rn = r.copy()
    rn[1000:1010] = np.nan
    G.assert_f64_close(orc.ewmst(ts, rn, 30.0), d["ewmst_nan_30.0"], rtol=1e-11, what="ewmst nan")
    for span in (2, 20, 500):
        G.assert_f64_close(orc.ewms(rn, span), d[f"ewms_{span}"], rtol=1e-11, what=f"ewms {span}")
    for win, smp in ((2, 1), (50, 1), (50, 0)):
        G.assert_f64_close(orc.realized_vol(rn, win, bool(smp)), d[f"rv_{win}_{smp}"], rtol=1e-12,
                           what=f"rv {win}")


def test_tick_size_golden(orc):
    d = G.load("tick_size")
    _, px, _, _ = G.synth_from(orc, d["synth"])
    assert orc.comp_price_tick_size(px) == float(d["synth_tick"])
    for c in G.cases(d):
        assert orc.comp_price_tick_size(d[f"{c}__px"]) == float(d[f"{c}__tick"]), c


def test_trade_size_golden(orc):
    d = G.load("trade_size")
    got = orc.comp_bar_trade_size_features(d["am"], d["theta"], d["ci"], 5.0)
    for k, g in zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"], got):
        np.testing.assert_array_equal(g, d[k], err_msg=k)     # pairwise sums, NumPy's percentile rule: bit-exact


def test_preprocess_loops_golden(orc):
    """merge_split_trades / comp_trade_side_vector (bar/utils.py:263-329, 26-46) incl. the head-relative 1e-8 rule."""
    d = G.load("preprocess")
    for name in ("a", "b", "c", "eps"):
        ts, px, am, ibm = (d[f"{name}__{k}"] for k in ("ts", "px", "am", "ibm"))
        for tag, flag in (("side", ibm), ("noside", None)):
            if f"{name}__{tag}_ts" not in d:
                continue
            got = orc.merge_split_trades(ts, px, am, flag)
            for g, k in zip(got, ("ts", "px", "am", "sd")):
                w = d[f"{name}__{tag}_{k}"]
                assert g.dtype == w.dtype, (name, tag, k)
                np.testing.assert_array_equal(g, w, err_msg=f"{name} {tag} {k}")
        np.testing.assert_array_equal(orc.comp_trade_side_vector(px), d[f"{name}__tickrule"])


def test_cusum_indexer_golden(orc):
    """_cusum_bar_indexer (logic.py:152-221): indices and the in-place sigma forward fill."""
    d = G.load("cusum")
    for name in ("ewm", "ewm_lowfloor", "const", "allnan", "floor"):
        fl, mult = d[f"{name}__params"]
        got, filled = orc._cusum_bar_indexer(d[f"{name}__ts"], d[f"{name}__px"], d[f"{name}__sigma"], fl, mult,
                                             return_sigma=True)
        np.testing.assert_array_equal(got, d[f"{name}__idx"], err_msg=name)
        np.testing.assert_array_equal(filled, d[f"{name}__sigma_filled"], err_msg=name)


def _vp_inputs(orc, d, name):
    ts, px, _, sd = G.synth_from(orc, d["synth"])
    am = d["amount"]
    interval, window, nbins, va = d[f"{name}__params"]
    clock, ci = orc._time_bar_indexer(ts, interval)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    off, flat, _ = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    return clock[1:], o[1], o[2], off, flat, window, (None if nbins < 0 else int(nbins)), va


def test_volume_profile_rolling_golden(orc):
    """volume_profile_rolling (feature/core/volume.py:403-456): bucketed / raw levels, wide and narrow windows."""
    d = G.load("volume_profile")
    for name in ("m1_w30", "m1_w5_nobins", "s10_w120_b5", "m1_w30_b200"):
        bts, hi, lo, off, flat, window, nbins, va = _vp_inputs(orc, d, name)
        got = orc.volume_profile_rolling(bts, hi, lo, off, flat["price_levels"], flat["buy_volumes"],
                                         flat["sell_volumes"], window, nbins, 0.01, va)
        for g, k in zip(got, ("poc", "hva", "lva", "pct")):
            assert g.dtype == d[f"{name}__{k}"].dtype
            np.testing.assert_array_equal(g, d[f"{name}__{k}"], err_msg=f"{name}:{k}")


# ---- BASELINE cfg 1: 10^7 ticks -> 1-minute bars made by the reference's own TimeBarKit (oracle/gen_cfg1.py) ---------
def test_cfg1_reference_timebars_ohlcv(orc):
    d = G.load("cfg1_reference_timebars")
    n = int(d["n_ohlcv"])
    ts, px, am, sd = orc.synth(int(d["seed"]), 0, n)
    clock, ci = orc._time_bar_indexer(ts, 60.0)
    np.testing.assert_array_equal(clock, d["close_ts"])
    np.testing.assert_array_equal(ci, d["close_indices"])
    assert len(ci) - 1 == 8331
    np.testing.assert_array_equal(d["ohlcv_index_ns"], clock[1:])                 # rows are labelled with the close edge
    got = dict(zip(["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"],
                   orc.comp_bar_ohlcv(px, am, ci)))
    assert list(d["ohlcv_columns"]) == ["open", "high", "low", "close", "volume", "trades", "median_trade_size", "vwap"]
    for k, g in got.items():
        w = d["ohlcv_col_" + k]
        assert g.dtype == w.dtype, k
        np.testing.assert_array_equal(g, w, err_msg=k)                            # sequential restatement: exact, vwap too


def test_cfg1_reference_timebars_flow(orc):
    d = G.load("cfg1_reference_timebars")
    n = int(d["n_flow"])
    ts, px, am, sd = orc.synth(int(d["seed"]), 0, n)
    _, ci = orc._time_bar_indexer(ts, 60.0)
    np.testing.assert_array_equal(ci, d["flow_close_indices"])
    names = [str(c) for c in d["dir_columns"]]
    assert len(names) == 14
    for name, g in zip(names, orc.comp_bar_directional_features(px, am, ci, sd)):   # the frame keeps the tuple's order
        np.testing.assert_array_equal(g, d["dir_col_" + name], err_msg=name)
    off, flat, bar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, d["flow_ohlcv_col_low"], d["flow_ohlcv_col_high"], 3.0)
    np.testing.assert_array_equal(np.diff(off), d["fp_n_levels"])
    for k, v in flat.items():
        np.testing.assert_array_equal(v.astype(d["fp_" + k].dtype), d["fp_" + k], err_msg=k)
    for k, v in bar.items():
        if k == "vp_skew":                                                        # rounding noise of a BLAS dot
            np.testing.assert_allclose(v, d["fp_" + k], atol=1e-6)
        else:
            np.testing.assert_array_equal(v, d["fp_" + k], err_msg=k)


def test_cfg3_reference_threshold_indexers(orc):
    """The reference's own sequential volume / dollar indexers (bar/logic.py:87-149) at 10^7 ticks of the bench stream and on a
    2*10^6-tick lognormal float64 tape (every addition rounds; the dollar carry never resets): the oracle's close indices."""
    d = G.load("cfg1_reference_timebars")
    ts, px, am, sd = orc.synth(int(d["seed"]), 0, int(d["n_ohlcv"]))
    np.testing.assert_array_equal(orc._volume_bar_indexer(am, float(d["cfg3_vthr"])), d["cfg3_volume_close_indices"])
    np.testing.assert_array_equal(orc._dollar_bar_indexer(px, am, float(d["cfg3_dthr"])), d["cfg3_dollar_close_indices"])
    lam, lpx = G.lognormal_tape(d)
    np.testing.assert_array_equal(orc._volume_bar_indexer(lam, float(d["cfg3_logn_vthr"])), d["cfg3_logn_volume_close_indices"])
    np.testing.assert_array_equal(orc._dollar_bar_indexer(lpx, lam, float(d["cfg3_logn_dthr"])), d["cfg3_logn_dollar_close_indices"])


def test_tick_level_chain_reference_vectors(orc):
    """comp_lagged_returns -> ewmst -> _cusum_bar_indexer made by the reference's own loops (oracle/gen_ticklevel_chain.py):
    returns bit-exact (one IEEE division / log), sigma to 1e-12 (libm exp), the CUSUM closes computed from the ORACLE's sigma
    identical."""
    d = G.load("ticklevel_chain_reference")
    ts, px, am, sd = orc.synth(int(d["seed"]), 0, int(d["n"]))
    k = int(d["step"])
    r = orc.comp_lagged_returns(ts, px, float(d["return_window_sec"]), True)
    sg = orc.ewmst(ts, r, float(d["half_life_sec"]))
    assert int(np.isnan(r).sum()) == int(d["returns_nan"]) and int(np.isnan(sg).sum()) == int(d["sigma_nan"])
    np.testing.assert_allclose(r[::k], d["returns_sampled"], rtol=1e-15, atol=0, equal_nan=True)
    np.testing.assert_allclose(sg[::k], d["sigma_sampled"], rtol=1e-12, atol=0, equal_nan=True)
    np.testing.assert_array_equal(orc._cusum_bar_indexer(ts, px, sg.copy(), float(d["sigma_floor"]), float(d["lambda_mult"])),
                                  d["cusum_close_indices"])


def test_oracle_bar_loops_do_not_depend_on_the_thread_count(orc, monkeypatch):
    """ORC_THREADS only splits the BAR loop (rows 5-7 are loops over independent bars in the reference); the arithmetic inside
    a bar is untouched, so 1 and 4 threads give the same bits -- the full-size GPU parity tests and bench.py's cpu_baseline run
    the oracle on all host cores."""
    ts, px, am, sd = orc.synth(42, 0, 300_000)
    am2 = (am * np.float32(1.2345678)).astype(np.float32)                 # inexact float32 sums as well
    _, ci = orc._time_bar_indexer(ts, 60.0)
    outs = []
    for threads in ("1", "4"):
        monkeypatch.setenv("ORC_THREADS", threads)
        res = []
        for a in (am, am2):
            o = orc.comp_bar_ohlcv(px, a, ci)
            d = orc.comp_bar_directional_features(px, a, ci, sd)
            off, flat, bar = orc.comp_bar_footprints_csr(px, a, ci, sd, 0.01, o[2], o[1], 3.0)
            res += list(o) + list(d) + [off] + list(flat.values()) + list(bar.values())
        outs.append(res)
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):                                       # base.py:719 through the threaded loop as well
        orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2] + 0.05, o[1], 3.0)


@pytest.mark.parametrize("prefix,ci_key,n_key", [("", "close_indices", "n"), ("s1_", "s1_close_indices", "s1_n")])
def test_oracle_on_float32_non_dyadic_amounts_against_reference_vectors(orc, prefix, ci_key, n_key):
    """The input class real data belongs to (float32 sizes whose sums round): vectors the REFERENCE made from float64 carriers of
    the float32 values -- its accumulators are then float64 as under Numba's typing (oracle/gen_f32amounts.py) -- against the
    oracle fed the float32 column itself."""
    d = G.load("f32_amounts_reference")
    n = int(d[n_key])
    ts, px, _, sd = orc.synth(42, 0, n)
    am = G.f32_amounts(d)[:n]
    assert am.dtype == np.float32
    _, ci = orc._time_bar_indexer(ts, 60.0 if prefix == "" else 1.0)
    np.testing.assert_array_equal(ci, d[ci_key])
    o = dict(zip(["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"], orc.comp_bar_ohlcv(px, am, ci)))
    dd = dict(zip(G.DIR_KEYS, orc.comp_bar_directional_features(px, am, ci, sd))) if prefix == "" else None
    off, flat, bar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o["low"], o["high"], 3.0)
    theta = d[prefix + "theta"] if prefix else d["theta"]
    np.testing.assert_array_equal(theta, o["median_trade_size"])
    t32 = dict(zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"], orc.comp_bar_trade_size_features(am, theta, ci, 5.0)))
    nd = G.check_f32_amount_vectors(d, prefix, n, ci_key, o, dd, (np.diff(off), flat, bar), t32, what=f"oracle {prefix or '1min'}")
    print(f"{prefix or '1min'}: imbalance flags differing between the float64 product (oracle, Numba typing) and the recorded "
          f"float32 product: {nd}")
    assert nd <= 2
    if prefix == "":
        # the float64 carrier through the trade-size reducer (NumPy float64 reductions on both sides)
        t64 = orc.comp_bar_trade_size_features(am.astype(np.float64), theta, ci, 5.0)
        for key, got in zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"], t64):
            np.testing.assert_array_equal(got, d["ts64_col_" + key], err_msg=key)


def test_pct_above_poc_nan_total(orc):
    """volume.py:378 `total_volume <= 0` does not catch a NaN total: the quotient (NaN) is returned."""
    pl = np.array([1, 2, 3, 4], np.int32)
    assert np.isnan(orc.calc_volume_percentage_above_poc(pl, np.array([1.0, np.nan, 2.0, 1.0], np.float32), 2))
    assert orc.calc_volume_percentage_above_poc(pl, np.zeros(4, np.float32), 2) == 0.0


def test_cot_is_numpy_argmax_with_nan_levels(orc):
    """base.py:829 `np.argmax(total_volumes)`: NumPy treats a NaN as the maximum and returns the FIRST one (a NaN size makes its
    level's sum NaN).  The oracle's loop is checked against np.argmax itself on its own level sums: NaN on the lowest level, on a
    middle one, on two levels, and none."""
    rng = np.random.default_rng(3)
    n = 4000
    px = 100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, n))
    sd = rng.choice(np.array([-1, 1], np.int8), n)
    ci = np.array([-1, 1999, n - 1], np.int64)
    for nan_at in ([], [17], [17, 2500], [0]):
        am = rng.lognormal(-1, 1, n).astype(np.float32)
        am[nan_at] = np.nan
        o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
        off, flat, bar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
        for b in range(2):
            lv = flat["price_levels"][off[b]:off[b + 1]]
            tv = flat["buy_volumes"][off[b]:off[b + 1]] + flat["sell_volumes"][off[b]:off[b + 1]]
            assert bar["cot_price_levels"][b] == lv[np.argmax(tv)], (nan_at, b)


@pytest.mark.parametrize("kind", G.TS_LENGTH_KINDS)
def test_trade_size_over_bar_lengths_against_reference_vectors(orc, kind):
    """oracle/gen_tradesize_lengths.py: the reference's comp_bar_trade_size_features (base.py:549-612) on float32 sizes, bars of 0 ..
    90 000 ticks on both sides of every schedule edge of the HIP path, lognormal / decimal-lot / dyadic sizes with a NaN size, a zero
    theta, an all-zero bar.  The oracle reproduces every column bit for bit."""
    d = G.load("trade_size_lengths_reference")
    am, theta, ci = G.tradesize_lengths_inputs(kind)
    np.testing.assert_array_equal(am[::997], d[kind + "_amount_check"])
    np.testing.assert_array_equal(ci, d[kind + "_close_indices"])
    np.testing.assert_array_equal(theta, d[kind + "_theta"])
    got = orc.comp_bar_trade_size_features(am, theta, ci, 5.0)
    for k, g in zip(G.TS_KEYS, got):
        if k == "pct_block" and am.dtype == np.float32:
            # `block_volume = 0.0; block_volume += amount` (base.py:599-603) is a float32 running sum in the recorded (pure-Python)
            # mode and a float64 one under Numba's typing, which the build follows: a sequential float32
            # sum over up to 90 000 sizes -- hence a tolerance for this column, here and in tests/_golden.py only
            np.testing.assert_allclose(g, d[kind + "_" + k], rtol=2e-5, atol=0, equal_nan=True, err_msg=f"{kind} {k}")
        else:
            np.testing.assert_array_equal(g, d[kind + "_" + k], err_msg=f"{kind} {k}")


def test_oracle_on_long_bars_against_reference_vectors(orc):
    """oracle/gen_longbars.py: the reference's four bar reducers on bars of 70 001 / 100 / 129 900 / 1 / 16 499 / 8 500 / 194 999 ticks
    (float32 lognormal sizes, float64 carriers where Numba's typing makes the accumulators float64) -- the lengths only the workgroup
    schedules of the HIP path serve, which until round 3 were pinned to the oracle alone."""
    d = G.load("long_bars_reference")
    n = int(d["lb_n"])
    ts, px, _, sd = orc.synth(42, 0, n)
    am = G.long_bars_amounts()
    np.testing.assert_array_equal(am[::9973], d["lb_amount_check"])
    ci = d["lb_close_indices"]
    o = dict(zip(["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"], orc.comp_bar_ohlcv(px, am, ci)))
    dd = dict(zip(G.DIR_KEYS, orc.comp_bar_directional_features(px, am, ci, sd)))
    off, flat, bar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o["low"], o["high"], 3.0)
    theta = d["lb_theta"]
    np.testing.assert_array_equal(theta, o["median_trade_size"])
    t32 = dict(zip(G.TS_KEYS, orc.comp_bar_trade_size_features(am, theta, ci, 5.0)))
    nd = G.check_f32_amount_vectors(d, "lb_", n, "lb_close_indices", o, dd, (np.diff(off), flat, bar), t32, what="oracle long bars")
    assert nd <= 2


def test_oracle_order_flow_near_tie_bar_with_a_nan_against_reference_vectors(orc):
    """tests/golden/nan_tie_longbar.npz (tools/gen_nan_tie_fixture.py): the reference's own order-flow columns of a 65 537-tick bar whose
    running dollar sum peaks 2.7e-12 below a float32 rounding boundary, with a NaN size later in the bar."""
    px, am, ci, sd, want = G.nan_tie_longbar()
    got = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    for k, g, w in zip(G.DIR_KEYS, got, want):
        np.testing.assert_array_equal(g, w, err_msg=k)
