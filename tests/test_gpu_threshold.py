"""GPU parity: _volume_bar_indexer / _dollar_bar_indexer close indices, bit-exact vs goldens and oracle."""
import os

import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu


def test_threshold_indexers_golden(orc):
    from finmlkit_amd.bar.logic import _dollar_bar_indexer, _volume_bar_indexer
    d = G.load("threshold_indexers")
    ts, px, am, sd = G.synth_from(orc, d["synth"])
    for k, want in d.items():
        kind, _, thr = k.partition("_")
        if kind == "vol32":
            got = _volume_bar_indexer(am, float(thr))
        elif kind == "dol32":
            got = _dollar_bar_indexer(px, am, float(thr))
        elif kind == "vol64":
            got = _volume_bar_indexer(d["r_am"], float(thr))
        elif kind == "dol64":
            got = _dollar_bar_indexer(d["r_px"], d["r_am"], float(thr))
        else:
            continue
        np.testing.assert_array_equal(got, want, err_msg=k)
        assert got.dtype == np.int64
    np.testing.assert_array_equal(_dollar_bar_indexer(d["big_px"], d["big_am"], 100.0), d["big_dol_100"])
    np.testing.assert_array_equal(_volume_bar_indexer(d["big_am"], 10.0), d["big_vol_10"])


@pytest.mark.parametrize("n,f64", [(2_000_000, False), (1_500_001, True), (1023, False), (1025, True), (1, False)])
def test_threshold_vs_oracle(orc, n, f64):
    from finmlkit_amd.bar.logic import _dollar_bar_indexer, _volume_bar_indexer
    ts, px, am, sd = orc.synth(17, 0, n)
    if f64:
        am = np.random.default_rng(6).lognormal(-1, 1.3, n)
    mean_v = float(np.mean(am))
    # 3000 ticks/bar: beyond the 2048-tick table span -> the 4096-tick tables; longer: wave-parallel chain walk
    for bar_ticks in (3, 50, 1200, 3000, 5000, 70_000, 400_000, 10**9):
        vthr = mean_v * bar_ticks
        np.testing.assert_array_equal(_volume_bar_indexer(am, vthr), orc._volume_bar_indexer(am, vthr),
                                      err_msg=f"vol {bar_ticks}")
        dthr = vthr * float(px[0])
        np.testing.assert_array_equal(_dollar_bar_indexer(px, am, dthr), orc._dollar_bar_indexer(px, am, dthr),
                                      err_msg=f"dol {bar_ticks}")


def test_threshold_serial_walk_forced(orc, monkeypatch):
    """FMK_THRESHOLD_SERIAL=1: both threshold indexers skip their parallel tiers (closed form + exact tier, jump tables) and run the
    serial walk -- the reference's loop, one wave -- on tapes the tiers would have served: the same closes, nothing uncertified."""
    from finmlkit_amd import engine
    from finmlkit_amd.bar.logic import _dollar_bar_indexer, _volume_bar_indexer
    monkeypatch.setenv("FMK_THRESHOLD_SERIAL", "1")
    n = 400_000
    ts, px, am, sd = orc.synth(21, 0, n)
    for am_k in (am, np.random.default_rng(2).lognormal(-1, 1.3, n)):
        mean_v = float(np.mean(am_k))
        for bar_ticks in (40, 1500):
            vthr = mean_v * bar_ticks
            np.testing.assert_array_equal(_volume_bar_indexer(am_k, vthr), orc._volume_bar_indexer(am_k, vthr))
            dthr = vthr * float(px[0])
            np.testing.assert_array_equal(_dollar_bar_indexer(px, am_k, dthr), orc._dollar_bar_indexer(px, am_k, dthr))
    t = engine.DeviceTrades.from_numpy(ts, px, am)
    t.volume_bar_index(float(np.mean(am)) * 300)
    assert t.last_uncertified == 0


def test_threshold_device_resident(orc):
    """Device-resident path + close_ts gather (kit.py:97-101) + OHLCV on volume bars."""
    from finmlkit_amd import engine
    n = 300_000
    t = engine.DeviceTrades.synth(n, seed=42)
    ts, px, am, sd = orc.synth(42, 0, n)
    ci = t.volume_bar_index(2048.0)
    want = orc._volume_bar_indexer(am, 2048.0)
    np.testing.assert_array_equal(ci.to_host(), want)
    assert t.last_uncertified == 0          # dyadic amounts: every decision certified
    np.testing.assert_array_equal(t.gather_ts(ci).to_host(), ts[want])
    got = engine.to_host(t.bar_ohlcv(ci))
    o = orc.comp_bar_ohlcv(px, am, want)
    np.testing.assert_array_equal(got["trades"], o[6])
    np.testing.assert_array_equal(got["median_trade_size"], o[7])


def test_threshold_knife_edge_is_exact_by_default(orc):
    """A threshold that divides the total exactly (thr = 3 * mean over n = 3k ticks) puts the last decision on a knife
    edge: in exact arithmetic the final tick closes a bar, the reference's float64 running sum -- drifted by ~1e-12 --
    does not (found by tools/fuzz_parity.py, seed 1 case 781).  The parallel indexer works on exact sums and REPORTS such
    decisions (n_uncertified); by default the library then redoes the input with the reference's own sequence of float64
    operations (k_threshold_exact), so the NumPy-facing functions are the oracle bit for bit.  With
    fmk_ctx_set_fast_threshold(1) the parallel result comes back with its count."""
    from finmlkit_amd import _ffi, engine
    from finmlkit_amd.bar.logic import _dollar_bar_indexer, _volume_bar_indexer
    rng = np.random.default_rng(5)
    hits = reported = 0
    for n in (8193, 3 * 2731, 3 * 4099, 3 * 700):
        px = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=n)), 0.01)
        for am in (rng.lognormal(-1, 1.2, size=n).astype(np.float32), rng.lognormal(-1, 1.2, size=n)):
            dthr = float(np.mean(am.astype(np.float64) * px)) * 3.0
            vthr = float(np.mean(am, dtype=np.float64)) * 3.0
            np.testing.assert_array_equal(_dollar_bar_indexer(px, am, dthr), orc._dollar_bar_indexer(px, am, dthr))
            np.testing.assert_array_equal(_volume_bar_indexer(am, vthr), orc._volume_bar_indexer(am, vthr))
            t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), px, am)
            ctx = _ffi.default_context()
            ctx.set_fast_threshold(True)
            try:
                fast = t.dollar_bar_index(dthr).to_host()
                unc = t.last_uncertified
            finally:
                ctx.set_fast_threshold(False)
            want = orc._dollar_bar_indexer(px, am, dthr)
            reported += unc > 0
            if not np.array_equal(fast, want):
                assert unc > 0, "the parallel result differs from the reference without reporting an uncertified decision"
                hits += 1
            assert np.array_equal(t.dollar_bar_index(dthr).to_host(), want) and t.last_uncertified == 0
    assert reported >= 4, f"the construction should put the last decision on the edge: {reported} of 8 inputs reported one"
    print("knife edge: uncertified reported for", reported, "of 8 inputs; the parallel result differed on", hits)


@pytest.mark.parametrize("seed,sigma,f64,L", [(1, 1.0, True, 3500), (3, 0.5, False, 8000), (4, 2.5, True, 6000),
                                             (5, 1.0, True, 20000), (6, 3.0, True, 12000), (7, 1.0, True, 40000)])
def test_volume_bars_of_thousands_of_ticks_continuous_amounts(orc, seed, sigma, f64, L):
    """Mean bar lengths between the LDS tables (<= 4096 ticks) and the chain walk: the global jump tables of
    fmk_volume.hip (k_vg_nxt / k_vg_level0).  Lognormal amounts, so prefix-sum differences are NOT the reference's
    sequential sums and the certification is live; whales (bars of one tick among the long ones); streams shorter than
    one table span; a threshold that is some tick's sequential running sum to the last bit (decision 1 on a knife edge:
    listed by the kernels, replayed by k_vol_verify, and redone serially when the replay disagrees)."""
    from finmlkit_amd.bar.logic import _volume_bar_indexer
    rng = np.random.default_rng(seed)
    for n in (1_200_000, 70_001, 5_000):
        am = rng.lognormal(0.0, sigma, n).astype(np.float64 if f64 else np.float32)
        if seed == 6:
            am[rng.integers(0, n, 20)] *= 5e4
        thr = float(am.astype(np.float64).mean()) * L
        edge = float(np.cumsum(am.astype(np.float64))[min(n - 1, L)])
        for t in (thr, edge):
            np.testing.assert_array_equal(_volume_bar_indexer(am, t), orc._volume_bar_indexer(am, t), err_msg=f"n={n} thr={t!r}")


def test_volume_long_bars_negative_amounts_take_the_serial_walk(orc):
    """Prefix sums must not decrease for any of the parallel tiers: a negative (or NaN) amount anywhere sends the input to
    the serial walk, whatever the bar length (the prefix pass of the long-bar tiers used not to look)."""
    from finmlkit_amd.bar.logic import _volume_bar_indexer
    rng = np.random.default_rng(11)
    for n, L in ((400_000, 5000), (400_000, 90_000), (3_000, 5000)):
        am = rng.lognormal(0.0, 1.0, n)
        am[n // 2] = -3.0 * am[n // 2] - 50.0
        thr = float(np.abs(am).mean()) * L
        np.testing.assert_array_equal(_volume_bar_indexer(am, thr), orc._volume_bar_indexer(am, thr), err_msg=f"n={n} L={L}")


def test_volume_exact_mode_certifies_on_the_chain_only(orc):
    """n_uncertified counts fragile decisions ON THE CHAIN OF CLOSES (those the result depends on), not over every tick's
    hypothetical bar -- and the default mode replays exactly those with the reference's sequential sum.  For continuous
    amounts the default therefore costs what the fast mode costs: it reports 0 after a confirmed replay."""
    from finmlkit_amd import _ffi, engine
    rng = np.random.default_rng(21)
    n = 2_000_000
    am = rng.lognormal(0.0, 1.0, n)
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), np.ones(n), am)
    ctx = _ffi.default_context()
    cs = np.cumsum(am)
    for L in (90, 900, 3000, 9000, 150_000):
        # thr = the sequential running sum at tick L: the first decision is an exact tie for the reference (it closes at L)
        for thr in (float(am.mean()) * L, float(cs[L])):
            want = orc._volume_bar_indexer(am, thr)
            ctx.set_fast_threshold(True)
            try:
                fast = t.volume_bar_index(thr).to_host()
                unc_fast = t.last_uncertified
            finally:
                ctx.set_fast_threshold(False)
            exact = t.volume_bar_index(thr).to_host()
            assert t.last_uncertified == 0
            np.testing.assert_array_equal(exact, want, err_msg=f"L={L} thr={thr!r}")
            if not np.array_equal(fast, want):
                assert unc_fast > 0, f"L={L}: the parallel result differs from the reference without a reported decision"
            if thr == float(cs[L]):
                assert unc_fast >= 1, f"L={L}: the tie at tick {L} was not reported"
            assert unc_fast < 50, f"L={L}: {unc_fast} fragile decisions on a chain of {len(want)} closes?"


@pytest.mark.parametrize("f64", [True, False])
def test_volume_decimal_lots_round_threshold(orc, f64):
    """Decimal lots with a round threshold: sums of 0.1, 0.2, ... hit the threshold EXACTLY in one summation order and
    miss it by an ulp in another (100 x 0.1 is 9.99999999999998 in tick order).  An exact tie of the parallel evaluation
    is therefore certain only for exactly-summable streams (multiples of 2^-20: the kernels check); here every such tie
    must be listed and replayed -- on all tiers (bar lengths 50 .. 200 000 ticks)."""
    from finmlkit_amd import _ffi, engine
    from finmlkit_amd.bar.logic import _volume_bar_indexer
    rng = np.random.default_rng(31)
    n = 1_000_000
    am = (rng.integers(1, 10, n) / 10.0).astype(np.float64 if f64 else np.float32)
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), np.ones(n), am)
    ctx = _ffi.default_context()
    reported = 0
    for thr in (25.0, 500.0, 1500.0, 2500.0, 10_000.0, 100_000.0):
        want = orc._volume_bar_indexer(am, thr)
        np.testing.assert_array_equal(_volume_bar_indexer(am, thr), want, err_msg=f"thr={thr}")
        ctx.set_fast_threshold(True)
        try:
            fast = t.volume_bar_index(thr).to_host()
            unc = t.last_uncertified
        finally:
            ctx.set_fast_threshold(False)
        reported += unc > 0
        if not np.array_equal(fast, want):
            assert unc > 0, f"thr={thr}: the parallel result differs from the reference without a reported decision"
    if f64:
        assert reported >= 3, f"exact ties of decimal lots should be reported: {reported} of 6 thresholds"


def test_volume_integer_lots_ties_are_certain(orc):
    """Integer (and dyadic) lots: every sum is exact in float64 in any order, an exact tie IS the reference's decision --
    nothing is listed although a large share of the closes are exact hits."""
    from finmlkit_amd import _ffi, engine
    rng = np.random.default_rng(32)
    n = 1_000_000
    am = (rng.integers(1, 9, n) * 0.25).astype(np.float32)
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), np.ones(n), am)
    ctx = _ffi.default_context()
    for thr in (10.0, 1000.0, 4000.0, 9000.0, 300_000.0):
        want = orc._volume_bar_indexer(am, thr)
        ctx.set_fast_threshold(True)
        try:
            fast = t.volume_bar_index(thr).to_host()
            unc = t.last_uncertified
        finally:
            ctx.set_fast_threshold(False)
        np.testing.assert_array_equal(fast, want, err_msg=f"thr={thr}")
        assert unc == 0, f"thr={thr}: {unc} decisions listed for an exactly-summable stream"


def test_volume_dyadic_lots_threshold_one_ulp_off_the_grid(orc):
    """Dyadic lots whose bars add up to EXACTLY the grid value just below the threshold: prefix + thr rounds to the grid, so
    the decision looks like an exact tie although the reference's `cum >= thr` is false (tools/fuzz_parity.py seed 778 case
    121).  Ties count as certain only when the threshold is on the amounts' grid too.  All tiers: short, 4096-tick and
    chain-walk bars."""
    from finmlkit_amd.bar.logic import _volume_bar_indexer
    rng = np.random.default_rng(33)
    n = 600_000
    am = (rng.integers(1, 65, n) * 0.125).astype(np.float32)
    for base in (40.0, 2840.5, 12_000.25, 90_000.0):
        for thr in (np.nextafter(base, np.inf), np.nextafter(base, -np.inf), base):
            got = _volume_bar_indexer(am, float(thr))
            want = orc._volume_bar_indexer(am, float(thr))
            np.testing.assert_array_equal(got, want, err_msg=f"thr={thr!r}")
    # the two neighbours of a grid value give different closes wherever a bar hits it exactly
    up, dn = orc._volume_bar_indexer(am, float(np.nextafter(2840.5, np.inf))), orc._volume_bar_indexer(am, 2840.5)
    assert not np.array_equal(up, dn)


def test_volume_decimal_lots_many_ties_stay_on_the_parallel_path(orc):
    """3e6 ticks of tenth lots, threshold 25: thousands of closes are exact ties of the parallel evaluation.  All of them are
    listed and replayed (the list holds 2^20 decisions; each replays one bar) -- the result is the oracle's and the call
    stays far below the serial walk's 15-22 ns per tick."""
    import time
    from finmlkit_amd import _ffi, engine
    rng = np.random.default_rng(33)
    n = 3_000_000
    am = rng.integers(1, 10, n) / 10.0
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), np.ones(n), am)
    ctx = _ffi.default_context()
    want = orc._volume_bar_indexer(am, 25.0)
    ctx.set_fast_threshold(True)
    try:
        t.volume_bar_index(25.0)
        listed = t.last_uncertified
    finally:
        ctx.set_fast_threshold(False)
    assert listed > 4096, f"the construction should tie on thousands of closes: {listed}"
    t.volume_bar_index(25.0); ctx.sync()
    t0 = time.perf_counter()
    got = t.volume_bar_index(25.0).to_host()
    dt = time.perf_counter() - t0
    np.testing.assert_array_equal(got, want)
    assert t.last_uncertified == 0
    assert dt < 0.02, f"{dt * 1e3:.1f} ms: the serial walk would take ~50 ms for {n} ticks"
    print(f"{listed} tied or near-tied decisions of {len(want) - 1} closes replayed; {dt * 1e3:.2f} ms")


@pytest.mark.parametrize("margin_scale", ["1", "1e5", "1e9"])
def test_dollar_exact_tier_forced(orc, monkeypatch, margin_scale):
    """csrc/fmk_dollar_exact.hip on inputs small enough for the oracle: forced (FMK_DL_FORCE_EXACT_TIER) so that it also runs
    where the closed form is already certain, with its margin widened so that from a few to ALL bars are replayed from the
    reconstructed float64 state.  Lognormal amounts (rounding in every add), decimal lots with a round threshold (exact ties
    in exact arithmetic that the reference's running sum misses: real flips, more than one round), dyadic amounts, bars of 3
    to 20 000 ticks, a threshold just below a power of two (the closing add lands in the next binade: state mod 4)."""
    from finmlkit_amd import _ffi, engine
    monkeypatch.setenv("FMK_DL_FORCE_EXACT_TIER", "1")
    monkeypatch.setenv("FMK_DL_MARGIN_SCALE", margin_scale)
    rng = np.random.default_rng(11)
    n = 400_000
    px = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=n)), 0.01)
    streams = {
        "lognormal64": rng.lognormal(-1.0, 1.0, n),
        "lognormal32": rng.lognormal(-1.0, 1.0, n).astype(np.float32),
        "tenths": rng.integers(1, 10, n) / 10.0,
        "dyadic": (rng.integers(1, 4097, n) * 2.0 ** -10).astype(np.float32),
    }
    for name, am in streams.items():
        mean = float(np.mean(am.astype(np.float64) * px))
        for thr in (mean * 3.0, mean * 57.3, mean * 20_000.0, 4096.0 - 1e-9, 1000.0):
            if float(np.max(am.astype(np.float64) * px)) >= thr:
                continue                                   # an increment >= thr: not this tier (serial walk)
            t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), px, am)
            got = t.dollar_bar_index(thr).to_host()
            assert t.last_uncertified == 0
            np.testing.assert_array_equal(got, orc._dollar_bar_indexer(px, am, thr), err_msg=f"{name} thr={thr!r}")


def test_dollar_exact_tier_end_of_stream(orc, monkeypatch):
    """The last decision on a knife edge: the replay may find one close more or one fewer than the closed form."""
    from finmlkit_amd import engine
    monkeypatch.setenv("FMK_DL_FORCE_EXACT_TIER", "1")
    rng = np.random.default_rng(3)
    for n in (3 * 2731, 3 * 4099, 30_000):
        px = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=n)), 0.01)
        for am in (rng.lognormal(-1, 1.2, size=n), rng.integers(1, 10, n) / 10.0):
            d = am * px
            for thr in (float(np.mean(d)) * 3.0, float(np.sum(d)) / 100.0, float(np.cumsum(d)[n - 1]) / 7.0):
                if float(np.max(d)) >= thr:
                    continue
                t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), px, am)
                np.testing.assert_array_equal(t.dollar_bar_index(thr).to_host(), orc._dollar_bar_indexer(px, am, thr))
                assert t.last_uncertified == 0


@pytest.mark.parametrize("whale", [3e5, 1e6])
def test_dollar_bars_after_a_whale_are_not_certified_by_the_tick_count_alone(orc, whale):
    """An increment of `whale` thresholds leaves a backlog: the reference closes a bar per tick while its running sum falls from
    ~whale * thr, and those adds round at THAT magnitude -- the drift of the later decisions is (whale)^2 ticks' worth, not the tick
    count's.  tools/fuzz_volume.py (seed 81003, dollar cases 792 / 1143) found closes right after a whale certified and wrong by one
    tick.  The default mode equals the reference's loop; the parallel mode must REPORT decisions now (it counted none before)."""
    from finmlkit_amd import _ffi, engine
    rng = np.random.default_rng(792)
    n = 600_000
    px = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=n)), 0.01)
    am = rng.integers(1, 20, n).astype(np.float32)
    am[150_000] *= np.float32(whale * 3.6)
    thr = 3594.9535145178074
    want = orc._dollar_bar_indexer(px, am, thr)
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), px, am)
    got = t.dollar_bar_index(thr).to_host()
    np.testing.assert_array_equal(got, want)
    assert t.last_uncertified == 0
    ctx = _ffi.default_context()
    ctx.set_fast_threshold(True)
    try:
        fast = t.dollar_bar_index(thr).to_host()
        unc = t.last_uncertified
    finally:
        ctx.set_fast_threshold(False)
    assert unc > 0, "the parallel mode certified every decision of a stream with a backlog"
    assert abs(len(fast) - len(want)) <= unc


def _fast_mode(t, thr):
    from finmlkit_amd import _ffi
    ctx = _ffi.default_context()
    ctx.set_fast_threshold(True)
    try:
        fast = t.volume_bar_index(thr).to_host()
        return fast, t.last_uncertified
    finally:
        ctx.set_fast_threshold(False)


def test_volume_chain_walk_reports_a_close_forced_onto_the_last_tick_of_a_block(orc):
    """tools/fuzz_volume.py seed 97015 case 2997 (the stream is the fixture: 36 689 lognormal float64 amounts, threshold = their
    correctly rounded total, which the reference's sequential sum stays one part in 1e15 below: no bar).  The chain walk picks the
    crossing block by double-double block totals and, when the plain in-block sums do not cross, closes on the block's last tick --
    without listing the decision: the parallel mode answered one bar and reported nothing."""
    from finmlkit_amd import engine
    d = np.load(os.path.join(G.GOLDEN_DIR, "volume_total_tie_case.npz"))
    a, thr = d["a"], float(d["thr"])
    n = len(a)
    want = orc._volume_bar_indexer(a, thr)
    assert len(want) == 1
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), np.ones(n), a)
    got = t.volume_bar_index(thr).to_host()
    np.testing.assert_array_equal(got, want)
    assert t.last_uncertified == 0
    fast, unc = _fast_mode(t, thr)
    assert np.array_equal(fast, want) or unc > 0, "the parallel mode differs from the reference and reports no fragile decision"


@pytest.mark.parametrize("seed", range(12))
def test_volume_threshold_at_the_rounded_total(orc, seed):
    """The same situation from seeds: thresholds at NumPy's (pairwise) total of the stream, where the exact sum, the pairwise sum and
    the reference's sequential sum disagree in the last bits; lengths that put the stream on the chain walk and on the global tables."""
    from finmlkit_amd import engine
    rng = np.random.default_rng(9700 + seed)
    n = int(rng.choice([5_000, 36_689, 70_001, 200_000]))
    a = rng.lognormal(0.0, float(rng.choice([0.1, 1.0])), n)
    k = int(rng.choice([1, 1, 2, 7]))                                # the total, or the sum of the first n / k ticks
    thr = float(a[: n // k].sum())
    want = orc._volume_bar_indexer(a, thr)
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), np.ones(n), a)
    got = t.volume_bar_index(thr).to_host()
    np.testing.assert_array_equal(got, want)
    assert t.last_uncertified == 0
    fast, unc = _fast_mode(t, thr)
    assert np.array_equal(fast, want) or unc > 0


@pytest.mark.parametrize("big", [1e9, 1e12])
def test_volume_fast_walk_next_to_an_amount_that_dwarfs_the_threshold(orc, big):
    """tools/fuzz_volume.py seed 97015 case 2356: a NaN amount sends the parallel mode to the chunked walk of fmk_threshold.hip, whose
    in-chunk sums are differences of chunk-wide prefixes -- after an amount of 1e12 the other ticks of that chunk (and the carry into the
    next one) are good to ulp(1e12) = 1e-4, with a threshold of 25.  Those decisions were off by a tick and not reported."""
    from finmlkit_amd import engine
    rng = np.random.default_rng(2356)
    n = 300_000
    a = rng.lognormal(0.0, 1.0, n).astype(np.float32).astype(np.float64)
    a[rng.integers(0, n, 6)] *= big
    a[int(0.8 * n)] = np.nan
    thr = 25.0
    want = orc._volume_bar_indexer(a, thr)
    t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), np.ones(n), a)
    got = t.volume_bar_index(thr).to_host()
    np.testing.assert_array_equal(got, want)
    assert t.last_uncertified == 0
    fast, unc = _fast_mode(t, thr)
    assert np.array_equal(fast, want) or unc > 0
    assert unc < len(want) // 4, "the magnitude term must stay local to the chunks next to the large amounts"


def test_volume_bar_of_small_trades_right_behind_a_block_trade(orc):
    """tools/fuzz_volume.py seed 5202 case 241 (round 5): tenth lots with a few trades of ~7e11 in 1.7e6 ticks, threshold 1000.  One
    block trade closes its bar, and the ~2000-tick bar of tenth lots that starts right behind it -- in the same 512-tick prefix
    block -- ends on a sum of 1000.0000055: the block-local prefixes behind 7e11 are good to 1e-4, the margin of the global-table
    and chain-walk tiers knew only about 1e-11 x the threshold, and the close came out one tick late, certified."""
    from finmlkit_amd import engine
    from tools import fuzz_case
    dist, a, px, thr = fuzz_case.regenerate(5202, 241, 3_000_000, "volume")
    assert dist == "decimal" and a.dtype == np.float64 and len(a) == 1738199 and thr == 1000.0
    want = orc._volume_bar_indexer(a, thr)
    t = engine.DeviceTrades.from_numpy(np.arange(len(a), dtype=np.int64), px, a)
    got = t.volume_bar_index(thr).to_host()
    np.testing.assert_array_equal(got, want)
    assert t.last_uncertified == 0
    fast, unc = _fast_mode(t, thr)
    assert np.array_equal(fast, want) or unc > 0


def _dollar_path():
    import ctypes as C
    from finmlkit_amd import _ffi
    p = C.c_int64()
    _ffi.lib().fmk_diag_dollar_last(C.byref(p))
    return p.value


@pytest.mark.parametrize("n,share,factor", [(100_000_000, 1e-4, 1000.0), (20_000_000, 1e-3, 8000.0)])
def test_dollar_bars_with_block_trades_stay_on_the_parallel_path(orc, monkeypatch, n, share, factor):
    """a tape with block trades (increments >= the threshold) used to leave the exact tier -- one fragile decision
    then cost the serial walk, ~20 s per 1e9 ticks.  1e8 synthetic ticks with 0.01 % of the sizes x 1000 (each ~1.2 thresholds: a
    backlog of one or two closes), and 2e7 ticks with 0.1 % x 8000 (backlogs of ~9 closes, overlapping now and then): the closes of the
    default exact mode against the sequential oracle, n_uncertified 0, answered by the exact tier's stretch walk (path 2)."""
    from finmlkit_amd import _ffi, engine
    from finmlkit_amd._ffi import DeviceArray
    monkeypatch.setenv("FMK_DL_FORCE_EXACT_TIER", "1")
    ctx = _ffi.default_context()
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
    am = t.amount.to_host()
    idx = np.random.default_rng(5).integers(0, n, int(n * share))
    am[idx] *= np.float32(factor)
    px = t.price.to_host()
    t2 = engine.DeviceTrades(ctx, t.ts, t.price, DeviceArray.from_host(ctx, am), t.side)
    thr = float((am[:2_000_000].astype(np.float64) * px[:2_000_000]).mean()) * 865.0 / (1 + share * factor)
    assert float((am[idx].astype(np.float64) * px[idx]).max()) >= thr          # the tape does hold block trades
    got = t2.dollar_bar_index(thr).to_host()
    assert t2.last_uncertified == 0
    assert _dollar_path() == 2
    want = orc._dollar_bar_indexer(px, am, thr)
    np.testing.assert_array_equal(got, want)
