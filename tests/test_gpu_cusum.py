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


def _last_tier():
    import ctypes as C
    from finmlkit_amd import _ffi
    from finmlkit_amd._ffi import c_i64
    t, o, s = c_i64(), c_i64(), c_i64()
    _ffi.lib().fmk_diag_cusum_last(C.byref(t), C.byref(o), C.byref(s))
    return t.value, o.value, s.value


def _last_segments():
    import ctypes as C
    from finmlkit_amd import _ffi
    from finmlkit_amd._ffi import c_i64
    k = c_i64()
    _ffi.lib().fmk_diag_cusum_segments(C.byref(k), None)
    return k.value


@pytest.mark.parametrize("n,vol,floor,mult,kind,same_ts", [
    (3_000_000, 2e-6, 5e-4, 2.0, "ewm", 0.25),        # the reference's default floor on a quiet tape: a close per ~1e5 ticks
    (3_000_000, 2e-6, 5e-4, 2.0, "const0", 0.0),      # sigma below the floor everywhere, no print blocks
    (1_000_000, 2e-5, 5e-4, 2.0, "ewm", 0.5),         # a close every few thousand ticks
    (300_000, 2e-4, 1e-6, 1.0, "ewm", 0.25),          # a close every few ticks: every chunk opened, dozens of restarts in each
    (300_001, 2e-4, 0.05, 2.0, "const0", 0.1),        # never reached: the walk opens nothing
    (4097, 2e-5, 5e-4, 2.0, "const0", 0.25),
])
@pytest.mark.parametrize("joint,segments", [(0, "128"), (0, "7"), (0, "1"), (1, "128")])
def test_cusum_chain_walk_vs_oracle(orc, monkeypatch, n, vol, floor, mult, kind, same_ts, joint, segments):
    """The chain walk for rarely reached thresholds (fmk_cusum_chain.hip), forced for every regime (no budget, no minimum
    size), as two independent side chains + merge -- each side in one piece or in segments that start where its state
    provably no longer depends on the past (k_cc_sync) -- and as the one joint walk: the same close indices as the
    sequential loop, and the tier must be the one that answered."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    # sample=50: two launches; the first ends inside a group of 64 chunks
    monkeypatch.setenv("FMK_CUSUM_CHAIN", f"2:joint={joint}:sample=50:segments={segments}")
    monkeypatch.setenv("FMK_CUSUM_CHAIN_MIN_CHUNKS", "2")
    ts, px = _stream(orc, n, 11, vol=vol, same_ts=same_ts)
    if kind == "ewm":
        r = orc.comp_lagged_returns(ts, px, 5.0, True)
        sigma = orc.ewmst(ts, r, 60.0)
        sigma[n // 3: n // 3 + 50] = np.nan
    else:
        sigma = np.full(n, 1e-7)
        sigma[:5] = np.nan                                     # the loop starts after the first valid sigma
    want, wfilled = orc._cusum_bar_indexer(ts, px, sigma, floor, mult, return_sigma=True)
    s = sigma.copy()
    got = _cusum_bar_indexer(ts, px, s, floor, mult)
    tier, opened, status = _last_tier()
    seg = _last_segments()
    print(f"n {n}: {len(want) - 1} closes, tier {tier}, chunks opened {opened}, status {status}, later segments {seg}")
    assert (tier, status) == (1, 0)
    if joint or segments == "1":
        assert seg == 0
    elif n >= 3_000_000:                                     # (a tape that closes every few hundred ticks has no such boundary)
        assert seg > 0, "no segment started although the tape has thousands of chunk boundaries"
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(s, wfilled)


@pytest.mark.parametrize("scale", ["1e9", "3e10"])
@pytest.mark.parametrize("joint", [0, 1])
def test_cusum_chain_walk_replays_what_its_margins_cannot_settle(orc, monkeypatch, scale, joint):
    """Decisions inside the margin are settled by the reference's own sequence of operations from the side's last reset
    (cc_replay).  Margins inflated by 1e9 / 3e10 put a share of the ticks near every close there -- also ticks that do NOT
    close, after which the walk has to go on as if nothing happened -- on the lattice tape of the bench round floors do it
    unaided (log-prices are multiples of a quantum: s == floor to the last bits)."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    monkeypatch.setenv("FMK_CUSUM_CHAIN", f"2:joint={joint}:sample=50")
    monkeypatch.setenv("FMK_CUSUM_CHAIN_MIN_CHUNKS", "2")
    monkeypatch.setenv("FMK_CUSUM_MARGIN_SCALE", scale)
    n = 1_000_000
    ts, px = _stream(orc, n, 19, vol=2e-5, same_ts=0.3)
    r = orc.comp_lagged_returns(ts, px, 5.0, True)
    sigma = orc.ewmst(ts, r, 60.0)
    want = orc._cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0)
    assert len(want) > 200
    got = _cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0)
    tier, opened, status = _last_tier()
    print(f"scale {scale}: {len(want) - 1} closes, tier {tier}, opened + events + replays {opened}, status {status}")
    assert (tier, status) == (1, 0)
    assert opened > 2.2 * len(want)                                  # replays happened (each counts, beside open + event)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("lead,hole", [(1_100_000, None), (7, 1_234_567), (0, None), (2_000_000, None)])
def test_cusum_sigma_is_filled_lazily(orc, monkeypatch, lead, hole):
    """sigma is not forward filled before the chain walk: its first valid index comes from the head of the array (k_ff_head:
    the first 2^20 entries, else the full pass), and a NaN after it is reported by the walk's summary pass, which then has the
    array filled and starts over.  Leading NaNs beyond the head, a hole far inside, none at all, all NaN: closes and the
    filled sigma as the reference leaves them (logic.py:178-189)."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    monkeypatch.setenv("FMK_CUSUM_CHAIN_MIN_CHUNKS", "2")
    n = 2_000_000
    ts, px = _stream(orc, n, 23, vol=2e-6, same_ts=0.25)
    sigma = np.full(n, 1e-7)
    sigma[:lead] = np.nan
    if hole is not None:
        sigma[hole: hole + 3] = np.nan
        sigma[hole - 1] = 4e-4                                     # the filled value matters: lam = 2 * 4e-4 there
    want, wfilled = orc._cusum_bar_indexer(ts, px, sigma, 5e-4, 2.0, return_sigma=True)
    for lazy in ("1", "0"):
        monkeypatch.setenv("FMK_CUSUM_LAZY_FILL", lazy)
        s = sigma.copy()
        got = _cusum_bar_indexer(ts, px, s, 5e-4, 2.0)
        np.testing.assert_array_equal(got, want, err_msg=f"lazy {lazy}")
        np.testing.assert_array_equal(s, wfilled, err_msg=f"lazy {lazy}")
        if lead < n:
            assert _last_tier()[0] == 1


def test_cusum_chain_walk_falls_back(orc, monkeypatch):
    """Decisions that block sums cannot settle at every tick (forced: margins scaled by 1e12), a non-finite return (a zero
    price) and a tape whose thresholds are reached often (the budget of opened chunks) all hand the call to the fixed point:
    same result."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    monkeypatch.setenv("FMK_CUSUM_CHAIN_MIN_CHUNKS", "2")
    n = 3_000_000
    ts, px = _stream(orc, n, 13, vol=1e-6, same_ts=0.25)
    sigma = np.full(n, 1e-7)
    want = orc._cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0)
    assert 5 < len(want) < 40
    monkeypatch.setenv("FMK_CUSUM_MARGIN_SCALE", "1e12")
    np.testing.assert_array_equal(_cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0), want)
    # (every tick is "inside the margin" now: each is settled by replaying the side from its last reset, which the budget
    #  of the sample phase ends after a hundred of them; a replay longer than 2^21 ticks would end it as "uncertain")
    assert _last_tier()[0] == 0 and _last_tier()[2] in (1, 2)
    monkeypatch.delenv("FMK_CUSUM_MARGIN_SCALE")
    np.testing.assert_array_equal(_cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0), want)
    assert _last_tier()[0] == 1
    # thresholds reached every few ticks: the sample's budget ends the walk
    want = orc._cusum_bar_indexer(ts, px, sigma.copy(), 1e-5, 2.0)
    np.testing.assert_array_equal(_cusum_bar_indexer(ts, px, sigma.copy(), 1e-5, 2.0), want)
    assert _last_tier()[0] == 0 and _last_tier()[2] == 1
    # a zero price: log(0 / p) = -inf, log(p / 0) = +inf
    px2 = px.copy()
    px2[n // 2] = 0.0
    with np.errstate(all="ignore"):
        want = orc._cusum_bar_indexer(ts, px2, sigma.copy(), 5e-4, 2.0)
    np.testing.assert_array_equal(_cusum_bar_indexer(ts, px2, sigma.copy(), 5e-4, 2.0), want)
    assert _last_tier()[0] == 0 and _last_tier()[2] == 3


def test_cusum_chain_walk_budget_of_a_segment(orc, monkeypatch):
    """A tape whose beginning is quiet (the estimate from the leading 2048 chunks says "walk") and whose last quarter closes
    every few dozen ticks: the segments there exhaust their budgets of opened sub-blocks + events, the tier reports it
    (status 1) and the fixed point answers -- same indices."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    monkeypatch.setenv("FMK_CUSUM_CHAIN_MIN_CHUNKS", "2")
    n = 6_000_000
    ts, _ = _stream(orc, n, 29, vol=1e-6, same_ts=0.25)
    rng = np.random.default_rng(29)
    lr = rng.normal(0, 1e-6, n)
    lr[4_500_000:] = rng.normal(0, 3e-4, n - 4_500_000)            # the last quarter: a threshold's worth every few ticks
    px = 100.0 * np.exp(np.cumsum(lr))
    sigma = np.full(n, 1e-7)
    want = orc._cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0)
    assert len(want) > 100_000
    got = _cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0)
    tier, opened, status = _last_tier()
    print(f"{len(want) - 1} closes, tier {tier}, opened {opened}, status {status}, later segments {_last_segments()}")
    assert (tier, status) == (0, 1) and _last_segments() > 0
    np.testing.assert_array_equal(got, want)


def test_cusum_chain_walk_positive_close_hides_negative(orc, monkeypatch):
    """`if s_pos >= lam ... elif s_neg <= -lam` (logic.py:214-219): inside a same-timestamp block the price falls by 3.5
    thresholds and recovers 2.2 of them on the block's last tick -- both sides are beyond their threshold there, only the
    positive one closes and resets, the negative side closes one tick later.  The two independent side chains put a close of
    each side on the same tick; the merge must notice and the joint walk must answer."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    monkeypatch.setenv("FMK_CUSUM_CHAIN_MIN_CHUNKS", "2")
    n = 600_000
    ts, px = _stream(orc, n, 17, vol=1e-6, same_ts=0.0)
    lr = np.diff(np.log(px), prepend=np.log(px[0]))
    for t in (200_000, 431_234):
        ts[t - 1] = ts[t - 2]; ts[t] = ts[t - 2]                  # ticks t-2, t-1 cannot close, tick t can
        assert ts[t + 1] > ts[t]
        lr[t - 1] = -3.5e-3; lr[t] = 2.2e-3; lr[t + 1] = 0.0
    px = 100.0 * np.exp(np.cumsum(lr))
    sigma = np.full(n, 1e-7)
    want = orc._cusum_bar_indexer(ts, px, sigma.copy(), 1e-3, 2.0)
    for t in (200_000, 431_234):
        assert t in want and t + 1 in want
    got = _cusum_bar_indexer(ts, px, sigma.copy(), 1e-3, 2.0)
    assert _last_tier()[0] == 1
    np.testing.assert_array_equal(got, want)


def _onepass():
    """(answered, fix-up launches, chunks pending after the first launch, chunks) of the last call (csrc/fmk_cusum_onepass.h)"""
    import ctypes as C
    from finmlkit_amd import _ffi
    from finmlkit_amd._ffi import c_i64
    v = [c_i64() for _ in range(4)]
    _ffi.lib().fmk_diag_cusum_onepass(*(C.byref(x) for x in v))
    return tuple(x.value for x in v)


@pytest.mark.parametrize("n,vol,floor,same_ts,sigma_kind", [
    (3_000_000, 2e-4, 2e-3, 0.25, "const"),           # a close every ~100 ticks, print blocks across chunk edges
    (1_000_000, 2e-4, 1e-6, 0.0, "const"),            # a close on nearly every tick: staging rows almost full
    (1_000_000, 2e-4, 1e-3, 0.5, "ewm"),              # sigma with a NaN prefix (first > 0) and a NaN hole (the lazy fill's redo)
    (1 + 3 * 4096 - 1, 2e-4, 2e-3, 0.25, "const"),    # the stream ends one tick before / on / one tick after a chunk edge
    (1 + 3 * 4096, 2e-4, 2e-3, 0.25, "const"),
    (1 + 3 * 4096 + 1, 2e-4, 2e-3, 0.25, "const"),
    (1 + 4096 + 512, 2e-4, 2e-3, 0.0, "const"),       # the second chunk is exactly one warm-up long
    (4097, 2e-4, 2e-3, 0.0, "const"), (4098, 2e-4, 2e-3, 0.0, "const"), (700, 2e-4, 2e-3, 0.0, "const")])
def test_cusum_onepass_vs_oracle(orc, monkeypatch, n, vol, floor, same_ts, sigma_kind):
    """The one-pass form of the dense regime (pass A with a warm-up, lockstep fix-up, emission) against the sequential oracle
    and against the fixed point it replaces, with the chain tier switched off so that it is the form that answers."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    monkeypatch.setenv("FMK_CUSUM_CHAIN", "0")
    ts, px = _stream(orc, n, 21, vol=vol, same_ts=same_ts)
    if sigma_kind == "ewm":
        r = orc.comp_lagged_returns(ts, px, 5.0, True)
        sigma = orc.ewmst(ts, r, 60.0)
        sigma[n // 2: n // 2 + 50] = np.nan
    else:
        sigma = np.full(n, floor / 4)
    want, wfilled = orc._cusum_bar_indexer(ts, px, sigma.copy(), floor, 2.0, return_sigma=True)
    s = sigma.copy()
    got = _cusum_bar_indexer(ts, px, s, floor, 2.0)
    used, launches, pending, chunks = _onepass()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(s, wfilled)
    assert used == 1 and chunks == -(-(n - 1 - int(want[0])) // 4096), (used, launches, pending, chunks)
    monkeypatch.setenv("FMK_CUSUM_ONEPASS", "0")
    s = sigma.copy()
    np.testing.assert_array_equal(_cusum_bar_indexer(ts, px, s, floor, 2.0), want)
    assert _onepass()[0] == 0


def test_cusum_onepass_leaves_a_tape_that_does_not_forget(orc, monkeypatch):
    """Thresholds that are never reached on a quiet tape: the clamps alone do not bring the walk from (0, 0) and the true
    walk together within the warm-up plus the first fix-up launch, so the form gives the call to the fixed point."""
    from finmlkit_amd.bar.logic import _cusum_bar_indexer
    monkeypatch.setenv("FMK_CUSUM_CHAIN", "0")
    n = 600_000
    ts, px = _stream(orc, n, 23, vol=1e-6, same_ts=0.1)
    sigma = np.full(n, 1e-7)
    want = orc._cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0)
    got = _cusum_bar_indexer(ts, px, sigma.copy(), 5e-4, 2.0)
    used, launches, pending, chunks = _onepass()
    np.testing.assert_array_equal(got, want)
    assert used == 0 and launches == 1 and pending > chunks // 4, (used, launches, pending, chunks)
