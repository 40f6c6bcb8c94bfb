"""GPU: comp_bar_ohlcv (incl. the median) across EVERY bar-length schedule of csrc/fmk_ohlcv.hip against the oracle -- one lane per
bar (<= 64 ticks), sixteen lanes per bar (65 .. 256), the wave-per-bar size classes (.. 1 344), the phased wave-per-bar classes
(.. 2 048 / 3 072 / 4 096 / 6 144), the workgroup per bar (.. 8 192), the generic kernels (.. 16 384) and the one-pass wide kernel
with the sample-bracket median beyond.  Bars of irregular lengths around each target, float32 lognormal sizes (every sum rounds),
a NaN size and a NaN first price in some bar, an empty bar; and the same bars inside a stream of very different bars (a bar's
result must not depend on its neighbours' lengths)."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu

KEYS = ["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"]
TARGETS = [20, 45, 70, 100, 200, 300, 700, 1200, 1400, 1900, 2100, 3000, 4000, 5000, 7000, 9000, 12000, 20000, 70000, 400000]


def _bars(rng, n, target):
    """close indices of bars of about `target` ticks (0.6x .. 1.4x), one empty bar, starting at -1"""
    lens = rng.integers(max(1, int(target * 0.6)), int(target * 1.4) + 2, size=n // max(target, 1) + 4)
    ci = np.concatenate([[-1], np.cumsum(lens) - 1])
    ci = ci[ci < n]
    if len(ci) > 6:
        ci = np.insert(ci, 5, ci[5])                                     # an empty bar
    if ci[-1] != n - 1:
        ci = np.append(ci, n - 1)
    return ci.astype(np.int64)


def _check(orc, t, engine, px, am, ci, what, median=True):
    got = engine.to_host(t.bar_ohlcv(engine.DeviceArray.from_host(t.ctx, ci), want_median=median))
    want = orc.comp_bar_ohlcv(px, am, ci, want_median=median)
    for k, w in zip(KEYS, want):
        if k == "median_trade_size" and not median:
            continue
        if k == "vwap":
            G.assert_f64_close(got[k], w, rtol=1e-9, what=f"{what}: vwap")
        else:
            np.testing.assert_array_equal(got[k], w, err_msg=f"{what}: {k}")
    return got


@pytest.mark.parametrize("target", TARGETS)
def test_every_bar_length_schedule(orc, target):
    from finmlkit_amd import engine
    rng = np.random.default_rng(target)
    n = 1_200_000 if target < 20000 else 3_000_000
    ts, px, _, sd = orc.synth(5, 0, n)
    am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    ci = _bars(rng, n, target)
    px = px.copy()
    if len(ci) > 12:
        am[int(ci[8]) + 1 + (int(ci[9]) - int(ci[8])) // 2] = np.nan      # a NaN size inside bar 8
        px[int(ci[10]) + 1] = np.nan                                      # a NaN FIRST price of bar 10
    t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
    _check(orc, t, engine, px, am, ci, f"bars of ~{target} ticks")
    _check(orc, t, engine, px, am, ci, f"bars of ~{target} ticks, no median", median=False)


def test_a_bar_gets_the_same_bits_in_any_stream(orc):
    """One stream made of sections of very different bar lengths, and each section alone: the mean bar length picks the schedule
    of the SHORT bars only, every longer bar goes by its own length -- results must be identical bar by bar."""
    from finmlkit_amd import engine
    rng = np.random.default_rng(77)
    n = 2_400_000
    ts, px, _, sd = orc.synth(6, 0, n)
    am = rng.lognormal(-1, 1.2, n).astype(np.float32)
    t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
    parts, at = [np.array([-1], dtype=np.int64)], 0
    for target, span in ((30, 100_000), (1500, 300_000), (150, 200_000), (2500, 300_000), (5000, 300_000), (7500, 300_000),
                         (12000, 300_000), (40000, 600_000)):
        c = _bars(rng, span, target)[1:] + at
        parts.append(c)
        at += span
    ci = np.concatenate(parts)
    whole = _check(orc, t, engine, px, am, ci, "mixed stream")
    # the long bars alone (as a sharded boundary launch would see them): same bits
    pos = 0
    for k in range(len(parts) - 1):
        lo, hi = pos, pos + len(parts[k + 1])
        sub = ci[lo:hi + 1]
        got = engine.to_host(t.bar_ohlcv(engine.DeviceArray.from_host(t.ctx, sub)))
        for key in KEYS:
            np.testing.assert_array_equal(got[key], whole[key][lo:hi], err_msg=f"section {k}: {key}")
        pos = hi


@pytest.mark.parametrize("gscale", ["0", "0.3"])
def test_wide_median_bracket_miss_takes_the_radix_select(orc, monkeypatch, gscale):
    """FMK_WIDE_MED_GSCALE shrinks the sample bracket of the one-pass wide kernel so that it misses for most bars: those go through
    the fallback list to the three-pass radix select -- np.median's bits either way."""
    from finmlkit_amd import engine
    monkeypatch.setenv("FMK_WIDE_MED_GSCALE", gscale)
    rng = np.random.default_rng(9)
    n = 2_000_000
    ts, px, _, sd = orc.synth(8, 0, n)
    am = np.where(rng.random(n) < 0.3, np.float32(0.001), rng.lognormal(-1, 1.2, n)).astype(np.float32)   # heavy ties too
    t = engine.DeviceTrades.from_numpy(ts, px, am, sd)
    _check(orc, t, engine, px, am, _bars(rng, n, 30000), f"wide bars, bracket scale {gscale}")
