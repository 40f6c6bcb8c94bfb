#!/usr/bin/env python3
"""Seeded differential test of the four bar reducers on LONG bars (the workgroup-per-bar schedules of round 3) against the oracle:
bar lengths around every threshold of those schedules (8 192, 16 384, 32 768, 65 536 ticks, -1 / 0 / +1) mixed with short, empty and
very long ones; amounts dyadic (few or many units: per-key totals below / above 2^24), full-mantissa float32, float64, with a NaN / a
negative size now and then; price grids coarse to fine (tens to thousands of levels per bar); sides with unsigned ticks.
    python tools/fuzz_longbars.py [cases] [seed]      exit code 1 on any difference"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import _golden as G

EDGE = [8191, 8192, 8193, 16383, 16384, 16385, 32767, 32768, 32769, 65535, 65536, 65537]
# mid=True: the bar lengths between the schedules of the short and the long end (the one-read trade-size kernels: one wave up to
# 1 920 ticks, 2 / 4 / 8 / 16 waves up to 3 824 / 7 648 / 15 840 / 16 384 -- two of np.sum's 8 192-element chunks beyond 8 192; the
# register classes of the medians and the footprints)
EDGE_MID = [128, 129, 256, 257, 1024, 1025, 1296, 1297, 1344, 1345, 1920, 1921, 2048, 2049, 3824, 3825, 4096, 4097, 7648, 7649, 8192,
            8193, 15840, 15841, 16384, 16385, 24032, 24033, 24576, 24577, 32224, 32225, 32768, 32769]


# mid="short": STREAMS of 17 000 .. 40 000 short bars -- what selects the lane-per-bar and sixteen-lanes-per-bar schedules and their
# hand-over lists (bars of 0 .. 64 / 65 .. 256 ticks, longer ones in between; the three-leaf lengths 249 .. 255 of NumPy's tree)
EDGE_SHORT = [0, 1, 7, 8, 9, 32, 33, 63, 64, 65, 127, 128, 129, 192, 193, 240, 248, 249, 250, 255, 256, 257, 300, 1344, 1345, 2048]


def case(rng, orc, pkg, k, mid=False):
    lens = []
    if mid == "short":
        nbars = int(rng.integers(17_000, 40_000))
        mean = float(rng.choice([12.0, 30.0, 60.0, 110.0, 200.0]))
        lens = np.minimum(rng.geometric(1.0 / mean, nbars) - 1, 3000).astype(np.int64)
        where = rng.integers(0, nbars, nbars // 50)
        lens[where] = rng.choice(EDGE_SHORT, len(where))
        if rng.random() < 0.6: lens = np.maximum(lens, 1)     # (no empty bar: the order-flow features are then defined for every bar and get compared)
        lens = [int(v) for v in lens]
    for _ in range(0 if mid == "short" else (int(rng.integers(2, 9)) if not mid else int(rng.integers(4, 24)))):
        u = rng.random()
        if mid:
            if u < 0.3: lens.append(int(rng.choice(EDGE_MID)))
            elif u < 0.4: lens.append(int(rng.integers(0, 300)))
            elif u < 0.8: lens.append(int(rng.integers(300, 8000)))
            else: lens.append(int(rng.integers(8000, 40_000)))
        elif u < 0.45: lens.append(int(rng.choice(EDGE)))
        elif u < 0.6: lens.append(int(rng.integers(0, 300)))
        elif u < 0.9: lens.append(int(rng.integers(8000, 140_000)))
        else: lens.append(int(rng.integers(140_000, 600_000)))
    first = -1 if rng.random() < 0.7 else int(rng.integers(0, 50))
    ci = (first + np.concatenate([[0], np.cumsum(lens)])).astype(np.int64)
    n = int(ci[-1]) + 1 + int(rng.integers(0, 40))
    step = float(rng.choice([0.5, 0.05, 0.01]))
    px = np.maximum(100.0 + step * np.cumsum(rng.integers(-2, 3, size=n)), step).astype(np.float64)
    kind = int(rng.integers(0, 5))
    if kind == 0: am = (rng.integers(1, 65, size=n) * 0.125).astype(np.float32)
    elif kind == 1: am = (rng.integers(1, 1 << 16, size=n) * 2.0 ** -4).astype(np.float32)
    elif kind == 2: am = rng.lognormal(-1, 1.2, size=n).astype(np.float32)
    elif kind == 3: am = rng.lognormal(-1, 1.2, size=n)
    else: am = (rng.integers(1, 4097, size=n) * 2.0 ** -10).astype(np.float32)
    if rng.random() < 0.1: am[int(rng.integers(0, n))] = np.nan
    if rng.random() < 0.1: am[int(rng.integers(0, n))] = -am[int(rng.integers(0, n))]
    sd = rng.choice(np.array([-1, 1, 1, -1, 0], dtype=np.int8), size=n)
    tick = float(rng.choice([step, step / 5, step * 4]))
    what = f"case {k}: lens {lens if len(lens) < 40 else str(lens[:12]) + ' ... ' + str(len(lens)) + ' bars'}, first {first}, step {step}, tick {tick}, amounts kind {kind} ({am.dtype})"
    o = orc.comp_bar_ohlcv(px, am, ci)
    got = pkg["base"].comp_bar_ohlcv(px, am, ci)
    for key, g, w in zip(["open", "high", "low", "close", "volume", "vwap", "trades", "median"], got, o):
        if key == "vwap": np.testing.assert_allclose(g, w, rtol=1e-9, err_msg=f"{what}: ohlcv {key}")
        else: np.testing.assert_array_equal(g, w, err_msg=f"{what}: ohlcv {key}")
    want = orc.comp_bar_directional_features(px, am, ci, sd, raise_on_zero_div=False)
    if not np.isnan(want[6]).any():
        got = pkg["base"].comp_bar_directional_features(px, am, ci, sd)
        for key, g, w in zip(G.DIR_KEYS, got, want):
            np.testing.assert_array_equal(g, w, err_msg=f"{what}: order flow {key}")
    woff, wflat, wbar = orc.comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], 3.0)
    if int(np.diff(woff).max(initial=0)) < (1 << 22):
        off, flat, bar = pkg["base"].comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], 3.0)
        np.testing.assert_array_equal(off, woff, err_msg=f"{what}: footprint offsets")
        for key in wflat: np.testing.assert_array_equal(flat[key], wflat[key], err_msg=f"{what}: footprints {key}")
        for key in wbar:
            if key == "vp_skew": np.testing.assert_allclose(bar[key], wbar[key], rtol=0, atol=1e-6, err_msg=f"{what}: footprints {key}")   # identically 0 in exact arithmetic (tests/test_gpu_features.py:_check_fp)
            else: np.testing.assert_array_equal(bar[key], wbar[key], err_msg=f"{what}: footprints {key}")
    if am.dtype == np.float32 or True:
        theta = np.full(len(ci) - 1, float(np.nanmedian(am)) if np.isfinite(np.nanmedian(am)) else 1.0)
        if rng.random() < 0.2: theta[int(rng.integers(0, len(theta)))] = 0.0
        want = orc.comp_bar_trade_size_features(am, theta, ci, 5.0)
        got = pkg["base"].comp_bar_trade_size_features(am, theta, ci, 5.0)
        for key, g, w in zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"], got, want):
            if am.dtype == np.float32: np.testing.assert_array_equal(g, w, err_msg=f"{what}: trade size {key}")
            else: np.testing.assert_allclose(g, w, rtol=2e-6, equal_nan=True, err_msg=f"{what}: trade size {key}")


def campaign(cases, seed, orc, verbose=True, mid=False):
    """-> list of failure messages"""
    from finmlkit_amd.bar import base
    pkg = {"base": base}
    rng = np.random.default_rng(seed)
    fails = []
    for k in range(cases):
        try:
            case(rng, orc, pkg, k, mid)
        except Exception as e:      # noqa: BLE001
            fails.append(f"seed {seed} {str(e)[:1500]}")
            if verbose:
                print(f"FAIL {fails[-1]}", flush=True)
                if not isinstance(e, AssertionError): traceback.print_exc()
    return fails


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from oracle import oracle as orc
    orc.build()
    mid = (sys.argv[3] if sys.argv[3] == "short" else sys.argv[3] == "mid") if len(sys.argv) > 3 else False
    fails = campaign(cases, seed, orc, mid=mid)
    print(f"{cases} {'short-bar-stream' if mid == 'short' else 'mid-length-bar' if mid else 'long-bar'} cases, seed {seed}: {len(fails)} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
