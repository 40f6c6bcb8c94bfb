"""GPU parity of the CUSUM bar indexer ("next" rank 3): parallel-in-time fixed point vs the reference goldens and the
sequential CPU oracle, plus CUSUMBarKit."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu


def test_cusum_golden():
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    d = G.load("cusum")
    for name in ("ewm", "ewm_lowfloor", "const", "allnan", "floor"):
        fl, mult = d[f"{name}__params"]
        sigma = d[f"{name}__sigma"].copy()
        got = _cusum_bar_indexer(d[f"{name}__ts"], d[f"{name}__px"], sigma, fl, mult)
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, d[f"{name}__idx"], err_msg=name)
        np.testing.assert_array_equal(sigma, d[f"{name}__sigma_filled"], err_msg=f"{name}: in-place fill")


def _stream(orc, n, seed, vol=2e-4, same_ts=0.25):
    ts, _, _, _ = orc.synth(seed, 0, n)
    rng = np.random.default_rng(seed)
    px = 100.0 * np.exp(np.cumsum(rng.normal(0, vol, n)))
    ts = ts.copy()
    blk = rng.random(n) < same_ts
    ts[1:][blk[1:]] = 0
    return np.maximum.accumulate(ts), px


@pytest.mark.parametrize("n,floor,mult,kind", [(1_000_000, 5e-4, 2.0, "ewm"), (1_000_000, 1e-6, 1.0, "ewm"),
                                               (500_000, 5e-4, 2.0, "const"), (300_000, 0.05, 2.0, "const"),
                                               (2049, 5e-4, 2.0, "ewm"), (2, 5e-4, 2.0, "const"), (1, 5e-4, 2.0, "const")])
def test_cusum_vs_oracle(orc, n, floor, mult, kind):
    """Thresholds from tight (a close every few ticks) to never reached (one long memory: many fixed-point rounds)."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    ts, px = _stream(orc, n, 5)
    if kind == "ewm" and n > 100:
        r = orc.comp_lagged_returns(ts, px, 5.0, True)
        sigma = orc.ewmst(ts, r, 60.0)
        sigma[n // 2: n // 2 + 50] = np.nan
    else:
        sigma = np.full(n, 1e-3)
    want, wfilled = orc._cusum_bar_indexer(ts, px, sigma, floor, mult, return_sigma=True)
    s = sigma.copy()
    got = _cusum_bar_indexer(ts, px, s, floor, mult)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(s, wfilled)


def test_cusum_rounds_and_kit(orc):
    import ctypes as C
    from finmlkit_amd import _ffi
    from finmlkit_amd._ffi import DeviceArray, c_f64, c_i64
    from finmlkit_amd.bar.data_model import TradesData
    from finmlkit_amd.bar.kit import CUSUMBarKit
    n = 400_000
    ts, px = _stream(orc, n, 9, same_ts=0.0)
    sigma = np.full(n, 1e-3)
    ctx = _ffi.default_context()
    m, rounds = c_i64(), c_i64()
    d_ts, d_px, d_sg = (DeviceArray.from_host(ctx, a) for a in (ts, px, sigma))      # keep the buffers alive
    ctx.call("fmk_cusum_bar_indexer_dev", d_ts.p, d_px.p, d_sg.p, c_i64(n), c_f64(5e-4), c_f64(2.0), None, c_i64(0),
             C.byref(m), C.byref(rounds))
    print("cusum rounds:", rounds.value, "closes:", m.value)
    want = orc._cusum_bar_indexer(ts, px, sigma, 5e-4, 2.0)
    assert m.value == len(want) and 2 <= rounds.value <= 64           # forgetting: a handful of rounds, not 196
    am = np.full(n, 0.5, dtype=np.float32)
    kit = CUSUMBarKit(TradesData(ts, px, am, np.arange(n), side=np.ones(n, np.int8)), sigma.copy(), 5e-4, 2.0)
    df = kit.build_ohlcv()
    np.testing.assert_array_equal(kit.bar_close_indices, want[1:])        # the open edge is excluded (base.py:116-125)
    assert len(df) == len(want) - 1
    o = orc.comp_bar_ohlcv(px, am, want)
    np.testing.assert_array_equal(df["close"].values, o[3])
    np.testing.assert_array_equal(kit.get_sigma(), sigma[want[1:]])
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    # logic.py:174 is the chained comparison len(prices) != len(sigma) != len(timestamps): it raises only when BOTH
    # inequalities hold (oracle/edge_sweep.py asked the reference) ...
    with pytest.raises(ValueError, match="same length"):
        _cusum_bar_indexer(ts, px, sigma[:-1].copy(), 5e-4, 2.0)
    # ... and with prices shorter than sigma = timestamps it works on the first len(prices) ticks
    np.testing.assert_array_equal(_cusum_bar_indexer(ts, px[:-1], sigma.copy(), 5e-4, 2.0),
                                  orc._cusum_bar_indexer(ts[:-1], px[:-1], sigma[:-1].copy(), 5e-4, 2.0))
