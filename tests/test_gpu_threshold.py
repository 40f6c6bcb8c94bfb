"""GPU parity: _volume_bar_indexer / _dollar_bar_indexer close indices, bit-exact vs goldens and oracle."""
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
