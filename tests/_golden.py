"""Helpers to read tests/golden/*.npz (written by oracle/gen_golden.py from the reference)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

OHLCV_KEYS = ["open", "high", "low", "close", "volume", "vwap", "trades", "median"]
DIR_KEYS = ["ticks_buy", "ticks_sell", "volume_buy", "volume_sell", "dollars_buy", "dollars_sell",
            "mean_spread", "max_spread", "cum_ticks_min", "cum_ticks_max", "cum_volumes_min",
            "cum_volumes_max", "cum_dollars_min", "cum_dollars_max"]
FP_LIST_KEYS = ["price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks",
                "buy_imbalances", "sell_imbalances"]
FP_BAR_KEYS = ["buy_imbalances_sum", "sell_imbalances_sum", "cot_price_levels",
               "imb_max_run_signed", "vp_skew", "vp_gini"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def cases(d):
    """Case names of a fixture whose keys look like '<case>__<field>'."""
    out = []
    for k in d:
        if "__" in k:
            c = k.split("__")[0]
            if c not in out:
                out.append(c)
    return out


def synth_from(orc, spec):
    seed, first, n, gap = (int(x) for x in spec)
    return orc.synth(seed, first, n, gap)


def reducer_stream(orc, d, case):
    """(prices, amounts, sides) of a reducers.npz case."""
    if case.startswith("syn_"):
        _, px, am, sd = synth_from(orc, d["syn__synth"])
    elif case.startswith("sparse_"):
        _, px, am, sd = synth_from(orc, d["sparse__synth"])
    else:
        px, am, sd = d["rnd__px"], d["rnd__am"], d["rnd__sd"]
    return px, am, sd


def assert_f64_close(got, want, rtol=1e-9, what=""):
    """The north-star float bar: <=1e-9 relative (NaN/inf positions must coincide)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    np.testing.assert_allclose(got, want, rtol=rtol, atol=0.0, equal_nan=True, err_msg=what)


def lognormal_tape(d):
    """(amounts float64, prices) of the lognormal tape of cfg1_reference_timebars.npz (oracle/gen_cfg1.py: same draws)."""
    rng = np.random.default_rng(int(d["cfg3_logn_seed"]))
    m = int(d["cfg3_logn_n"])
    lam = rng.lognormal(-1.0, 1.2, m)
    lpx = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=m)), 0.01)
    return lam, lpx


def f32_amounts(d):
    """The float32 NON-dyadic amount column of f32_amounts_reference.npz (oracle/gen_f32amounts.py: same draws), checked against
    the stored sample so that a NumPy whose generator changed fails here and not in a parity assertion."""
    am = np.random.default_rng(int(d["seed"])).lognormal(-1.0, 1.2, int(d["n"])).astype(np.float32)
    np.testing.assert_array_equal(am[::9973], d["amount_check"])
    return am


def check_f32_amount_vectors(d, prefix, n, ci_key, ohlcv, directional, footprints, trade_size32, *, what):
    """Compare one implementation's outputs (dicts of host arrays; `footprints` = (n_levels, flat dict, bar dict)) with the
    reference-made vectors of f32_amounts_reference.npz (`prefix` "" = one-minute bars of 1e6 ticks, "s1_" = one-second bars of
    1e5 ticks).  -> number of imbalance flags that differ (float32 vs float64 product, see oracle/gen_f32amounts.py)."""
    nb = len(d[ci_key]) - 1
    for key in ("open", "high", "low", "close", "volume", "trades", "median_trade_size"):
        want = d[prefix + "ohlcv_col_" + key]
        got = ohlcv[key][:nb]
        assert got.dtype == want.dtype, (what, key, got.dtype, want.dtype)
        np.testing.assert_array_equal(got, want, err_msg=f"{what}: {key}")
    assert_f64_close(ohlcv["vwap"][:nb], d[prefix + "ohlcv_col_vwap"], rtol=1e-9, what=f"{what}: vwap")
    if directional is not None:
        for name in (str(c) for c in d[prefix + "dir_columns"]):
            mine = {"cum_volume_min": "cum_volumes_min", "cum_volume_max": "cum_volumes_max"}.get(name, name)
            got, want = directional[mine][:nb], d[prefix + "dir_col_" + name]
            assert got.dtype == want.dtype, (what, name)
            if name in ("mean_spread", "max_spread"):                       # bar 0: wrap-around tick prices[-1]: the array's last
                got, want = got[1:], want[1:]
            if got.dtype == np.float32:
                # float64 sums rounded ONCE to float32: a reassociated float64 sum differs from the sequential one by ~1e-16
                # relative, which moves the float32 rounding only on a tie -- bit-identical in practice, and asserted so
                np.testing.assert_array_equal(got, want, err_msg=f"{what}: {name}")
            else:
                np.testing.assert_array_equal(got, want, err_msg=f"{what}: {name}")
    n_flag_diff = 0
    if footprints is not None:
        nlev, flat, bar = footprints
        np.testing.assert_array_equal(nlev[:nb], d[prefix + "fp_n_levels"], err_msg=f"{what}: levels per bar")
        nl = int(d[prefix + "fp_n_levels"].sum())
        for key in ("price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks"):
            want = d[prefix + "fp_" + key]
            np.testing.assert_array_equal(flat[key][:nl].astype(want.dtype), want, err_msg=f"{what}: {key}")
        for key in ("buy_imbalances", "sell_imbalances"):
            n_flag_diff += int((flat[key][:nl].astype(bool) != d[prefix + "fp_" + key].astype(bool)).sum())
        np.testing.assert_array_equal(bar["cot_price_levels"][:nb], d[prefix + "fp_cot_price_levels"], err_msg=f"{what}: cot")
        np.testing.assert_array_equal(bar["vp_gini"][:nb], d[prefix + "fp_vp_gini"], err_msg=f"{what}: vp_gini")
        np.testing.assert_allclose(bar["vp_skew"][:nb], d[prefix + "fp_vp_skew"], atol=1e-6, err_msg=f"{what}: vp_skew")
        if n_flag_diff == 0:
            for key in ("buy_imbalances_sum", "sell_imbalances_sum", "imb_max_run_signed"):
                np.testing.assert_array_equal(bar[key][:nb], d[prefix + "fp_" + key], err_msg=f"{what}: {key}")
    if trade_size32 is not None:
        for key in ("mean_size_rel", "size_95_rel", "pct_block", "size_gini"):
            want = d[prefix + "ts32_" + key]
            got = trade_size32[key][:nb]
            assert got.dtype == want.dtype == np.float32, (what, key)
            if key == "pct_block":
                # base.py:599-603 `block_volume = 0.0; block_volume += amount` is a float32 running sum in the recorded mode
                # (python float + np.float32 -> np.float32, NEP 50) and a float64 one under Numba's typing, which the build
                # follows for every scalar accumulator (typed-vs-recorded table): a float32 sequential sum
                # over ~1200 sizes carries ~sqrt(n) * 6e-8 relative error, hence a tolerance HERE and only here; the float64
                # carrier run below pins the float64 accumulation itself
                np.testing.assert_allclose(got, want, rtol=4e-6, atol=0, equal_nan=True, err_msg=f"{what}: trade-size {key}")
            else:
                np.testing.assert_array_equal(got, want, err_msg=f"{what}: trade-size {key}")
    return n_flag_diff


# ---- trade-size features over the bar-length axis (oracle/gen_tradesize_lengths.py made the expected columns with the reference) ----
TS_LENGTH_KINDS = ["lognormal", "lots", "dyadic", "lognormal64"]
TS_KEYS = ["mean_size_rel", "size_95_rel", "pct_block", "size_gini"]
# both sides of every edge between the schedules of the HIP path -- as they stood when the fixture was first made and as they are now
# (64 / 65 lanes -> rows, 128 / 129 rows -> one wave, the tree shapes at 1 024 / 1 025 and 1 296 / 1 297, 1 920 / 1 921 one wave -> two
# or one with five levels, 3 824 / 3 825, 7 648 / 7 649, 8 192 / 8 193 np.sum's second chunk, 15 840 / 15 841, 16 384 / 16 385 -> the
# sub-tree workgroup, 32 768 / 32 769 its sample-bracket percentile, 65 536 / 65 537) and lengths between
TS_LENGTHS = [1, 7, 8, 20, 63, 64, 65, 100, 128, 129, 200, 256, 257, 300, 600, 1023, 1024, 1025, 1200, 1296, 1297, 1800, 1920, 1921,
              2400, 3000, 3824, 3825, 5000, 7648, 7649, 12000, 15296, 15297, 24000, 30592, 30593, 32768, 32769, 50000, 65536, 65537,
              0, 90000, 1343, 1344, 1345, 2047, 2048, 2049, 4096, 4097, 8192, 8193, 8400, 15840, 15841, 16384, 16385, 24032, 24033, 24576,
              24577, 32224, 32225]


def tradesize_lengths_inputs(kind):
    """-> (amounts: float32, float64 for "lognormal64"; theta float64[B]; close indices int64[B+1]) of the `kind` stream; deterministic"""
    seed = {"lognormal": 11, "lots": 12, "dyadic": 13, "lognormal64": 14}[kind]
    rng = np.random.default_rng(seed)
    lens = list(TS_LENGTHS)
    rng.shuffle(lens)
    ci = np.cumsum([-1] + lens).astype(np.int64)
    n = int(ci[-1]) + 1
    if kind == "lognormal64":       # float64 sizes: np.sum's chunks and trees in float64, the block volume added in tick order
        am = rng.lognormal(-1.0, 1.2, n)
    elif kind == "lognormal":
        am = rng.lognormal(-1.0, 1.2, n).astype(np.float32)
    elif kind == "lots":            # decimal lot sizes: heavy ties, none of them a dyadic number
        am = (np.round(rng.lognormal(-1.0, 1.5, n), 2) + 0.01).astype(np.float32)
    else:
        am = (rng.integers(1, 4097, n) * 2.0 ** -10).astype(np.float32)
    nb = len(lens)
    theta = np.full(nb, float(np.median(am)))
    theta[3] = 0.0                                           # base.py:586-587: a NaN row
    j = int(np.argmax(np.array(lens) == 3000))               # one NaN size in the 3 000-tick bar, an all-zero 600-tick bar
    am[int(ci[j]) + 11] = np.nan
    z = int(np.argmax(np.array(lens) == 600))
    am[int(ci[z]) + 1:int(ci[z + 1]) + 1] = 0.0
    return am, theta, ci


# ---- the four bar reducers on long bars (oracle/gen_longbars.py made the expected columns with the reference) ----
LONG_BARS_N = 420_000
LONG_BARS_CUTS = [-1, 70_000, 70_100, 200_000, 200_001, 216_500, 225_000, 419_999]


def long_bars_amounts():
    return np.random.default_rng(4242).lognormal(-1.0, 1.2, LONG_BARS_N).astype(np.float32)


def nan_tie_longbar():
    """-> (prices, amounts float32, close_idx, sides, the oracle's 14 order-flow columns) of tests/golden/nan_tie_longbar.npz"""
    d = load("nan_tie_longbar")
    px = np.maximum(100.0 + 0.05 * np.cumsum(d["steps"].astype(np.int64)), 0.05)
    units = d["units"].astype(np.int64)
    am = (units * 2.0 ** -10).astype(np.float32)
    am[units == 0] = np.nan
    return px, am, d["close_idx"], d["sides"], tuple(d[f"want_{k}"] for k in DIR_KEYS)
