"""GPU parity of the first "next" row: comp_bar_trade_size_features."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu

KEYS = ["mean_size_rel", "size_95_rel", "pct_block", "size_gini"]


def test_trade_size_golden_f64():
    """float64 amounts: NumPy's pairwise trees, percentile rule and the tick-ordered block sum reproduced -> bit-exact."""
    from finmlkit_amd.bar.base import comp_bar_trade_size_features
    d = G.load("trade_size")
    got = comp_bar_trade_size_features(d["am"], d["theta"], d["ci"], 5.0)
    for k, g in zip(KEYS, got):
        assert g.dtype == np.float32
        np.testing.assert_array_equal(g, d[k], err_msg=k)
    assert np.isnan(got[0][3]) and np.isnan(got[3][3])          # theta == 0 -> NaN row


@pytest.mark.parametrize("n,interval", [(300_000, 60.0), (300_000, 1.0), (200_000, 3600.0)])
def test_trade_size_vs_oracle(orc, n, interval):
    from finmlkit_amd.bar.base import comp_bar_trade_size_features
    ts, px, am, sd = orc.synth(19, 0, n)
    am64 = np.random.default_rng(2).lognormal(-1, 1.2, n)
    _, ci = orc._time_bar_indexer(ts, interval)
    theta = np.full(len(ci) - 1, float(np.median(am64)))
    want = orc.comp_bar_trade_size_features(am64, theta, ci, 5.0)
    got = comp_bar_trade_size_features(am64, theta, ci, 5.0)
    for k, g, w in zip(KEYS, got, want):
        np.testing.assert_array_equal(g, w, err_msg=f"{k} iv={interval}")
    # float32 amounts (what TradesData's merge produces): the reference's mean / total / Gini sums are NumPy pairwise float32
    # sums and its percentile interpolates in float32; the kernel follows the same trees (fmk_pairwise.h) -> bit-identical
    theta32 = np.full(len(ci) - 1, float(np.median(am)))
    want = orc.comp_bar_trade_size_features(am, theta32, ci, 5.0)
    got = comp_bar_trade_size_features(am, theta32, ci, 5.0)
    for k, g, w in zip(KEYS, got, want):
        np.testing.assert_array_equal(g, w, err_msg=k)


@pytest.mark.parametrize("interval", [150.0, 300.0, 1800.0, 12000.0])
def test_trade_size_long_float32_bars(orc, interval):
    """float32 bars beyond the register classes (more than 2048 ticks): the percentile comes from the workgroup-per-bar radix
    select (k_ts_p95_long, 256 threads up to 8192 ticks, 1024 beyond), the two pairwise sums from fmk_pairwise_big -- lognormal
    sizes (every key distinct), a NaN in one bar, one bar whose close index lies beyond the array (Python slice semantics: the
    old path), against the oracle bit for bit."""
    from finmlkit_amd.bar.base import comp_bar_trade_size_features
    n = 700_000
    ts, px, am, sd = orc.synth(23, 0, n)
    am32 = np.random.default_rng(5).lognormal(-1, 1.2, n).astype(np.float32)
    _, ci = orc._time_bar_indexer(ts, interval)
    ci = ci.copy()
    assert len(ci) >= 3 and np.diff(ci).max() > 2048
    am32[int(ci[1]) + 7] = np.nan
    ci[-1] = n + 5                                               # amounts[start:end + 1] clamps
    theta = np.full(len(ci) - 1, float(np.nanmedian(am32)))
    want = orc.comp_bar_trade_size_features(am32, theta, ci, 5.0)
    got = comp_bar_trade_size_features(am32, theta, ci, 5.0)
    for k, g, w in zip(KEYS, got, want):
        np.testing.assert_array_equal(g, w, err_msg=f"{k} iv={interval}")


def test_trade_size_kit_and_errors(orc):
    import pandas as pd
    from finmlkit_amd.bar.base import comp_bar_trade_size_features
    from finmlkit_amd.bar.data_model import TradesData
    from finmlkit_amd.bar.kit import TickBarKit
    n = 50_000
    ts, px, am, sd = orc.synth(4, 0, n)
    am64 = am.astype(np.float64) * 1.37
    kit = TickBarKit(TradesData(ts, px, am64, np.arange(n), side=sd), 500)
    ci = orc._tick_bar_indexer(ts, 500)
    theta = np.full(len(ci) - 1, 1.0)
    df = kit.build_trade_size_features(theta, theta_mult=3.0)
    assert list(df.columns) == KEYS and len(df) == len(ci) - 1
    want = orc.comp_bar_trade_size_features(am64, theta, ci, 3.0)
    for k, w in zip(KEYS, want):
        np.testing.assert_array_equal(df[k].values, w, err_msg=k)
    with pytest.raises(ValueError, match="Theta should match"):
        comp_bar_trade_size_features(am64, theta[:-1], ci, 3.0)
    with pytest.raises(ValueError, match="Theta should match"):
        kit.build_trade_size_features(theta[:-1])


@pytest.mark.parametrize("interval,amounts", [(1.0, "dyadic"), (1.0, "lognormal32"), (2.5, "lognormal32"), (0.3, "lognormal32"),
                                              (1.0, "nan"), (400.0, "ties"), (10.0, "lognormal32"), (10.0, "nan"),
                                              (7.0, "dyadic"), (5.0, "tiny"), (10.0, "blocks")])
@pytest.mark.parametrize("mode", ["2", "0", "3"])
def test_trade_size_one_lane_per_bar(orc, monkeypatch, interval, amounts, mode):
    """Short float32 bars through the lane-per-bar schedule (FMK_TS_LANES=2 forces it whatever the number of bars; 3 = sixteen
    lanes per bar, bars up to 256 ticks; 0 = the wave-per-bar kernel alone): NumPy's pairwise leaf, float32 percentile and the float64 block sum evaluated by ONE lane per bar
    give the oracle's bits -- bars of 0 .. 64 ticks, longer ones and irregular close indices through the leftover list, theta == 0
    rows, NaN sizes, heavy ties; and the two schedules agree with each other."""
    from finmlkit_amd import engine
    monkeypatch.setenv("FMK_TS_LANES", mode)
    n = 120_000
    ts, px, am, sd = orc.synth(29, 0, n, orc.SPARSE_GAP_MOD if interval == 400.0 else orc.DENSE_GAP_MOD)
    rng = np.random.default_rng(11)
    if amounts != "dyadic":
        am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    if amounts == "nan":
        am[rng.integers(0, n, 300)] = np.nan
    if amounts == "ties":
        am = np.where(rng.random(n) < 0.6, np.float32(0.001), am).astype(np.float32)
    if amounts == "tiny":           # subnormal shares a / total (the sixteen-lane kernel's float64-product shares fall back to the division)
        am = (am * np.float32(1e-30)).astype(np.float32)
        am[rng.integers(0, n, 2000)] = np.float32(3e12)
        am[rng.integers(0, n, 2000)] = np.float32(1e-44)
    if amounts == "blocks":         # sizes above the block threshold in some bars only
        am[rng.integers(0, n, 400)] *= np.float32(300.0)
    _, ci = orc._time_bar_indexer(ts, interval)
    ci = ci.copy()
    nb = len(ci) - 1
    # a few long bars inside the stream (merge neighbours), an end index past the column (slice clamp)
    keep = np.ones(len(ci), bool)
    keep[200:215] = False
    keep[100:130] = False
    keep[1000:1002] = False
    if interval == 400.0:
        keep[300:360] = False
    ci = ci[keep]
    ci[-1] = n + 3
    nb = len(ci) - 1
    theta = np.full(nb, float(np.nanmedian(am)))
    theta[::97] = 0.0
    want = orc.comp_bar_trade_size_features(am, theta, ci, 5.0)
    t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
    got = t.bar_trade_size(engine.DeviceArray.from_host(t.ctx, ci), theta, 5.0)
    for k, w in zip(KEYS, want):
        np.testing.assert_array_equal(got[k], w, err_msg=f"{k} iv={interval} {amounts} mode={mode}")
    assert np.diff(ci).max() > 64 and np.median(np.diff(ci)) < 256
    assert interval != 400.0 or (np.diff(ci) == 0).any()


@pytest.mark.parametrize("case", ["edges", "huge", "zeros", "theta0", "dyadic"])
def test_trade_size_workgroup_per_bar(orc, case):
    """Bars of more than 32 768 ticks, float32 amounts (k_bar_trade_size_wide: the top of NumPy's pairwise tree cut into sub-trees
    that sixteen waves evaluate, added in the recursion's order; the percentile by radix select up to 65 536 ticks, from a sample
    bracket beyond): bars of 32 768 (still the wave kernel) / 32 769 / 40 000 / 65 536 / 65 537 / 100 003 ticks next to short
    ones; a bar of 2.3e6 ticks (sub-trees of 65 536 elements); all-zero amounts (total 0: pct and gini stay NaN); a zero theta;
    dyadic amounts (4 096 distinct sizes: heavy ties around the bracket).  Against the oracle bit for bit."""
    from finmlkit_amd.bar.base import comp_bar_trade_size_features
    rng = np.random.default_rng(41)
    if case == "huge":
        n = 2_600_000
        cuts = [-1, 2_300_000, 2_300_100, n - 1]
    else:
        lens = [100, 8192, 8193, 32_768, 3, 32_769, 65_537, 0, 100_003, 40_000, 65_536]
        cuts = [int(c) for c in np.cumsum([-1] + lens)]
        n = cuts[-1] + 50
    am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    if case == "zeros":
        am[cuts[5] + 1:cuts[6] + 1] = 0.0
    if case == "dyadic":
        am = (rng.integers(1, 4097, n) * 2.0 ** -10).astype(np.float32)
    ci = np.array(cuts, dtype=np.int64)
    theta = np.full(len(ci) - 1, float(np.median(am)))
    if case == "theta0":
        theta[2] = 0.0
        theta[6] = 0.0
    want = orc.comp_bar_trade_size_features(am, theta, ci, 5.0)
    got = comp_bar_trade_size_features(am, theta, ci, 5.0)
    for k, g, w in zip(KEYS, got, want):
        np.testing.assert_array_equal(g, w, err_msg=f"{k} ({case})")


@pytest.mark.parametrize("span", ["wave", "workgroup"])
@pytest.mark.parametrize("case", ["lognormal", "dyadic", "constant", "nan", "inf", "signed", "zeros", "tiny", "theta0", "few_values"])
def test_trade_size_one_read_wave_kernel(orc, case, span):
    """Regular float32 bars of 129 .. 1 920 ticks (k_bar_trade_size_mid: the bar read once into the registers of np.sum's leaf
    accumulators, both pairwise trees folded by DPP shifts, the share a / total as a float64 product, the percentile searched in the
    float domain and finished on a compacted register): every bar length around the kernel's ends (128 / 129, 1 920 / 1 921), the
    sizes where the tree changes shape (1 024 / 1 025: a ninth leaf; 1 296 / 1 297: leaves beyond 95 elements = the sixteen-term
    instantiation; 1 928 / 1 929 stay with the three-pass kernel) and random lengths between; sizes with heavy ties, all equal, NaN,
    +-inf, both signs, zeros of both signs, tiny sizes next to a large one (subnormal shares: the division fallback), two distinct
    values only.  span "workgroup": bars of 1 921 .. 16 384 ticks (k_bar_trade_size_wg: 2, 4, 8 or 16 waves per bar, each on the
    sub-tree at the end of its path through the top of np.sum's tree -- of each of its two 8 192-element chunks beyond 8 192 ticks --,
    the bar-wide values exchanged through LDS) around every edge between the wave counts (3 824 / 3 825, 7 648 / 7 649, 8 192 / 8 193,
    15 840 / 15 841, 16 384 / 16 385).  Against the oracle bit for bit."""
    from finmlkit_amd.bar.base import comp_bar_trade_size_features
    rng = np.random.default_rng(77)
    lens = [128, 129, 130, 136, 137, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1031, 1200, 1279, 1280, 1281, 1295, 1296, 1297,
            1344, 1500, 1919, 1920, 1921, 1928, 1929, 64, 1, 0] + [int(v) for v in rng.integers(129, 1921, 120)]
    if span == "workgroup":
        lens = [1921, 1929, 2000, 2048, 2049, 2400, 3600, 3823, 3824, 3825, 4096, 5000, 7648, 7649, 8192, 8193, 8200, 10105, 12000, 15840,
                15841, 16000, 16384, 16385, 20000, 24032, 24033, 24576, 24577, 32224, 32225, 32768, 32769, 100, 0] + [int(v) for v in rng.integers(1921, 32769, 14)]
    cuts = np.cumsum([-1] + lens)
    n = int(cuts[-1]) + 10
    am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    if case == "dyadic":
        am = (rng.integers(1, 4097, n) * 2.0 ** -10).astype(np.float32)
    elif case == "constant":
        am[:] = np.float32(0.37)
    elif case == "few_values":
        am = rng.choice(np.array([0.001, 2.5], dtype=np.float32), n)
    elif case == "nan":
        am[rng.integers(0, n, 25)] = np.nan
    elif case == "inf":
        am[rng.integers(0, n, 12)] = np.inf
        am[rng.integers(0, n, 12)] = -np.inf
    elif case == "signed":
        am *= rng.choice(np.array([-1.0, 1.0], dtype=np.float32), n)
    elif case == "zeros":
        z = rng.random(n)
        am[z < 0.5] = 0.0
        am[z < 0.2] = -0.0
        am[cuts[5] + 1:cuts[6] + 1] = 0.0
        am[cuts[7] + 1:cuts[8] + 1] = -0.0
    elif case == "tiny":
        am = (am * np.float32(1e-30)).astype(np.float32)
        am[rng.integers(0, n, 200)] = np.float32(3e12)
        am[rng.integers(0, n, 200)] = np.float32(1e-44)
    ci = cuts.astype(np.int64)
    theta = np.full(len(ci) - 1, float(np.nanmedian(np.abs(am[np.isfinite(am)]))) or 1.0)
    if case == "theta0":
        theta[::7] = 0.0
        theta[3] = np.nan
    with np.errstate(all="ignore"):
        want = orc.comp_bar_trade_size_features(am, theta, ci, 5.0)
    got = comp_bar_trade_size_features(am, theta, ci, 5.0)
    for k, g, w in zip(KEYS, got, want):
        np.testing.assert_array_equal(g, w, err_msg=f"{k} ({case})")


@pytest.mark.parametrize("kind", G.TS_LENGTH_KINDS)
def test_trade_size_over_bar_lengths_against_reference_vectors(kind):
    """The HIP path against the REFERENCE's own outputs (tests/golden/trade_size_lengths_reference.npz, made by
    oracle/gen_tradesize_lengths.py with finmlkit's comp_bar_trade_size_features): float32 sizes, bars on both sides of every edge
    between the seven schedules (one lane / sixteen lanes / one wave reading the bar once / 2, 4, 8, 16 waves / the sub-tree
    workgroup with 4 and 16 waves, radix-select and sample-bracket percentile), bit for bit."""
    from finmlkit_amd.bar.base import comp_bar_trade_size_features
    d = G.load("trade_size_lengths_reference")
    am, theta, ci = G.tradesize_lengths_inputs(kind)
    np.testing.assert_array_equal(am[::997], d[kind + "_amount_check"])
    got = comp_bar_trade_size_features(am, theta, ci, 5.0)
    for k, g in zip(G.TS_KEYS, got):
        if k == "pct_block" and am.dtype == np.float32:
            # `block_volume = 0.0; block_volume += amount` (base.py:599-603) is a float32 running sum in the recorded (pure-Python)
            # mode and a float64 one under Numba's typing, which the build follows: a sequential float32
            # sum over up to 90 000 sizes -- hence a tolerance for this column, here and in tests/_golden.py only
            np.testing.assert_allclose(g, d[kind + "_" + k], rtol=2e-5, atol=0, equal_nan=True, err_msg=f"{kind} {k}")
        else:
            np.testing.assert_array_equal(g, d[kind + "_" + k], err_msg=f"{kind} {k}")


@pytest.mark.parametrize("mode", ["3", "2"])
def test_trade_size_rows_schedule_leaves_three_leaf_bars(orc, monkeypatch, mode):
    """Bars of 249 .. 255 ticks have THREE leaves in NumPy's tree (the right half, 129 .. 135 elements, splits once more): the
    sixteen-lanes-per-bar schedule (two leaves at most) must hand them on.  tools/fuzz_parity.py seed 42001 case 11160 found them
    summed as two leaves (an ulp in mean_size_rel and pct_block); every length from 240 to 260 here, bit for bit."""
    from finmlkit_amd.bar.base import comp_bar_trade_size_features
    monkeypatch.setenv("FMK_TS_LANES", mode)
    rng = np.random.default_rng(42001)
    lens = list(range(240, 261)) * 6 + [int(v) for v in rng.integers(1, 300, 200)]
    rng.shuffle(lens)
    ci = np.cumsum([-1] + lens).astype(np.int64)
    n = int(ci[-1]) + 1
    am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    theta = np.full(len(lens), float(np.median(am)))
    want = orc.comp_bar_trade_size_features(am, theta, ci, 3.0)
    got = comp_bar_trade_size_features(am, theta, ci, 3.0)
    for k, g, w in zip(KEYS, got, want):
        np.testing.assert_array_equal(g, w, err_msg=f"{k} mode={mode}")
