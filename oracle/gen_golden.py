#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself.

Runs only in the build container (needs /root/reference).  The reference is
imported in the pure-Python mode its CI pins (NUMBA_DISABLE_JIT=1,
/root/reference/.github/workflows/ci.yml:36-39) through the tiny import shim
in oracle/shim (numba / dotenv are absent from the image).  The fixtures are
DATA: inputs (or the seed that regenerates them through oracle.synth) and the
reference's outputs.  No reference source is copied.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Inputs are chosen so that the reference's typed (Numba) semantics and its
pure-Python semantics coincide: amount columns are float64, or float32 with
dyadic values whose partial sums are exact (see oracle/fmk_oracle.c header).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)
os.environ["NUMBA_DISABLE_JIT"] = "1"

import numpy as np  # noqa: E402

from finmlkit.bar import base as rbase  # noqa: E402
from finmlkit.bar import logic as rlogic  # noqa: E402
from finmlkit.bar import utils as rutils  # noqa: E402
from finmlkit.feature.core import utils as rfutils  # noqa: E402
from finmlkit.feature.core import volatility as rvol  # noqa: E402

from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, d):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: {len(d)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def random_stream(seed, n, f64_amounts=True, zero_sides=False):
    """Non-dyadic random stream: stored verbatim in the fixture."""
    rng = np.random.default_rng(seed)
    ts = 1_700_000_000_000_000_000 + np.cumsum(rng.integers(1, 400_000_000, n)).astype(np.int64)
    k = 250_000 + np.cumsum(rng.integers(-2, 3, n))
    px = k * 0.01
    am = rng.lognormal(-2.0, 1.5, n)
    am = am if f64_amounts else am.astype(np.float32)
    sd = rng.choice(np.array([-1, 1], dtype=np.int8), n)
    if zero_sides:
        sd[rng.random(n) < 0.05] = 0
    return ts, px, am, sd


# ------------------------------------------------------------------ time indexer
def gen_time_indexer():
    d = {}
    cases = []
    # hand-sized streams in the style of the reference's own tests (small ts => trailing bar)
    cases.append(("small_a", np.array([999_999_999, 1_000_000_000, 2_000_000_000, 3_000_000_000,
                                       4_000_000_000, 5_000_000_000, 5_999_999_999, 6_100_000_000,
                                       7_000_000_000], dtype=np.int64), 2.0))
    cases.append(("small_gap", np.array([1_000_000_000, 2_000_000_000, 5_000_000_000, 6_000_000_000],
                                        dtype=np.int64), 2.0))
    cases.append(("single", np.array([1_500_000_000], dtype=np.int64), 1.0))
    cases.append(("on_edges", np.arange(0, 10_000_000_000, 1_000_000_000, dtype=np.int64), 1.0))
    cases.append(("dups", np.array([10, 10, 10, 2_000_000_000, 2_000_000_000, 2_000_000_001],
                                   dtype=np.int64), 1.0))
    cases.append(("half_second", np.array([100_000_000, 600_000_000, 1_100_000_000, 1_600_000_000,
                                           2_100_000_000], dtype=np.int64), 0.5))
    # epoch scale: ts[0] just below an edge (float64 rounding of ts[0] crosses the edge)
    e = 1_700_000_040_000_000_000
    cases.append(("epoch_round_up", np.array([e - 100, e - 50, e + 5, e + 70_000_000_000],
                                             dtype=np.int64), 60.0))
    cases.append(("epoch_on_edge", np.array([e, e + 1, e + 60_000_000_000, e + 60_000_000_001],
                                            dtype=np.int64), 60.0))
    for name, ts, iv in cases:
        clock, idx = rlogic._time_bar_indexer(ts, iv)
        d[f"{name}__ts"] = ts
        d[f"{name}__interval"] = np.float64(iv)
        d[f"{name}__clock"] = clock
        d[f"{name}__idx"] = idx
    # synthetic streams, regenerated from the seed in the tests
    for name, seed, n, gap, iv in [("dense60", 42, 20_000, orc.DENSE_GAP_MOD, 60.0),
                                   ("dense1", 42, 20_000, orc.DENSE_GAP_MOD, 1.0),
                                   ("dense3600", 7, 50_000, orc.DENSE_GAP_MOD, 3600.0),
                                   ("sparse60", 42, 5_000, orc.SPARSE_GAP_MOD, 60.0),
                                   ("sparse900", 3, 5_000, orc.SPARSE_GAP_MOD, 900.0)]:
        ts, _, _, _ = orc.synth(seed, 0, n, gap)
        clock, idx = rlogic._time_bar_indexer(ts, iv)
        d[f"{name}__synth"] = np.array([seed, 0, n, gap], dtype=np.int64)
        d[f"{name}__interval"] = np.float64(iv)
        d[f"{name}__clock"] = clock
        d[f"{name}__idx"] = idx
    save("time_indexer", d)


# ------------------------------------------------------------------ tick/volume/dollar
def gen_threshold_indexers():
    d = {}
    ts, px, am, sd = orc.synth(42, 0, 30_000)
    d["synth"] = np.array([42, 0, 30_000, orc.DENSE_GAP_MOD], dtype=np.int64)
    for thr in (1, 2, 7, 100, 1000, 40_000):
        d[f"tick_{thr}"] = np.array(rlogic._tick_bar_indexer(ts, thr), dtype=np.int64)
    for thr in (0.5, 3.0, 100.0, 2048.0, 1e9):
        d[f"vol32_{thr}"] = np.array(rlogic._volume_bar_indexer(am, thr), dtype=np.int64)
    for thr in (5e3, 2.5e6, 1e8, 1e15):
        d[f"dol32_{thr}"] = np.array(rlogic._dollar_bar_indexer(px, am, thr), dtype=np.int64)
    # non-dyadic float64 amounts (sequential float64 accumulation is order-sensitive here)
    rts, rpx, ram, rsd = random_stream(11, 20_000, f64_amounts=True)
    d["r_px"] = rpx
    d["r_am"] = ram
    for thr in (0.3, 5.0, 250.0):
        d[f"vol64_{thr}"] = np.array(rlogic._volume_bar_indexer(ram, thr), dtype=np.int64)
    for thr in (700.0, 1e5, 3e6):
        d[f"dol64_{thr}"] = np.array(rlogic._dollar_bar_indexer(rpx, ram, thr), dtype=np.int64)
    # one oversized tick (several thresholds in one trade) for the dollar carry rule
    p2 = np.array([10.0, 10.0, 10.0, 10.0, 10.0, 10.0, 10.0, 10.0])
    v2 = np.array([1.0, 1.0, 35.0, 0.1, 0.1, 0.1, 0.1, 9.0])
    d["big_px"] = p2
    d["big_am"] = v2
    d["big_dol_100"] = np.array(rlogic._dollar_bar_indexer(p2, v2, 100.0), dtype=np.int64)
    d["big_vol_10"] = np.array(rlogic._volume_bar_indexer(v2, 10.0), dtype=np.int64)
    save("threshold_indexers", d)


# ------------------------------------------------------------------ reducers
OHLCV_KEYS = ["open", "high", "low", "close", "volume", "vwap", "trades", "median"]
DIR_KEYS = ["ticks_buy", "ticks_sell", "volume_buy", "volume_sell", "dollars_buy", "dollars_sell",
            "mean_spread", "max_spread", "cum_ticks_min", "cum_ticks_max", "cum_volumes_min",
            "cum_volumes_max", "cum_dollars_min", "cum_dollars_max"]
FP_LIST_KEYS = ["price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks",
                "buy_imbalances", "sell_imbalances"]
FP_BAR_KEYS = ["buy_imbalances_sum", "sell_imbalances_sum", "cot_price_levels",
               "imb_max_run_signed", "vp_skew", "vp_gini"]


def gen_reducers():
    d = {}
    cases = []
    # (name, stream, close indices)
    ts, px, am, sd = orc.synth(42, 0, 20_000)
    d["syn__synth"] = np.array([42, 0, 20_000, orc.DENSE_GAP_MOD], dtype=np.int64)
    for iv in (60.0, 1.0):
        _, ci = rlogic._time_bar_indexer(ts, iv)
        cases.append((f"syn_t{int(iv)}", (px, am, sd), ci))
    cases.append(("syn_tick100", (px, am, sd), np.array(rlogic._tick_bar_indexer(ts, 100), dtype=np.int64)))
    cases.append(("syn_vol2048", (px, am, sd), np.array(rlogic._volume_bar_indexer(am, 2048.0), dtype=np.int64)))
    # sparse stream (empty bars, first index -1): OHLCV only has defined semantics there
    ts2, px2, am2, sd2 = orc.synth(42, 0, 5_000, orc.SPARSE_GAP_MOD)
    d["sparse__synth"] = np.array([42, 0, 5_000, orc.SPARSE_GAP_MOD], dtype=np.int64)
    _, ci2 = rlogic._time_bar_indexer(ts2, 60.0)
    # random float64 stream stored verbatim
    rts, rpx, ram, rsd = random_stream(5, 6_000, f64_amounts=True, zero_sides=True)
    d["rnd__ts"], d["rnd__px"], d["rnd__am"], d["rnd__sd"] = rts, rpx, ram, rsd
    _, rci = rlogic._time_bar_indexer(rts, 120.0)
    # drop empty bars for the directional golden (ZeroDivisionError in the reference)
    keep = np.concatenate([[True], np.diff(rci) > 0])
    rci_ne = rci[keep]
    cases.append(("rnd_t120", (rpx, ram, rsd), rci_ne))
    rci_tick = np.array(rlogic._tick_bar_indexer(rts, 37), dtype=np.int64)
    cases.append(("rnd_tick37", (rpx, ram, rsd), rci_tick))

    for name, (p, a, s), ci in cases:
        d[f"{name}__ci"] = ci
        o = rbase.comp_bar_ohlcv(p, a, ci)
        for k, v in zip(OHLCV_KEYS, o):
            d[f"{name}__ohlcv_{k}"] = np.asarray(v)
        nonempty = bool(np.all(np.diff(ci) > 0))
        if nonempty:
            dr = rbase.comp_bar_directional_features(p, a, ci, s.astype(np.int8))
            for k, v in zip(DIR_KEYS, dr):
                d[f"{name}__dir_{k}"] = np.asarray(v)
        tick = 0.01
        fp = rbase.comp_bar_footprints(p, a, ci, s.astype(np.int8), tick, o[2], o[1], 3.0)
        off = np.concatenate([[0], np.cumsum([len(x) for x in fp[0]])]).astype(np.int64)
        d[f"{name}__fp_offsets"] = off
        for k, lst in zip(FP_LIST_KEYS, fp[:7]):
            d[f"{name}__fp_{k}"] = np.concatenate([np.asarray(x) for x in lst]) if len(lst) else np.zeros(0)
        for k, v in zip(FP_BAR_KEYS, fp[7:]):
            d[f"{name}__fp_{k}"] = np.asarray(v)
    # sparse: OHLCV + footprints (empty bars included)
    d["sparse_t60__ci"] = ci2
    o2 = rbase.comp_bar_ohlcv(px2, am2, ci2)
    for k, v in zip(OHLCV_KEYS, o2):
        d[f"sparse_t60__ohlcv_{k}"] = np.asarray(v)
    save("reducers", d)


def gen_footprint_features():
    """comp_footprint_features on random level profiles (non-exact float32 sums: pins the
    pairwise float32 summation order of vp_gini)."""
    d = {}
    rng = np.random.default_rng(123)
    for i, L in enumerate([1, 2, 3, 7, 8, 9, 33, 100, 128, 129, 300]):
        lv = np.arange(1000, 1000 + L, dtype=np.int32)
        b = (rng.lognormal(0, 1.2, L)).astype(np.float32)
        s = (rng.lognormal(0, 1.2, L)).astype(np.float32)
        b[rng.random(L) < 0.15] = 0
        s[rng.random(L) < 0.15] = 0
        bi, si, run, cot, sk, gi = rbase.comp_footprint_features(lv, b, s, 1.5)
        d[f"c{i}__lv"], d[f"c{i}__b"], d[f"c{i}__s"] = lv, b, s
        d[f"c{i}__bi"], d[f"c{i}__si"] = bi, si
        d[f"c{i}__scalars"] = np.array([run, cot, sk, gi], dtype=np.float64)
    save("footprint_features", d)


def gen_trade_size():
    d = {}
    rts, rpx, ram, rsd = random_stream(9, 4_000, f64_amounts=True)
    ci = np.array(rlogic._tick_bar_indexer(rts, 53), dtype=np.int64)
    theta = np.full(len(ci) - 1, np.median(ram))
    theta[3] = 0.0
    out = rbase.comp_bar_trade_size_features(ram, theta, ci, 5.0)
    d["am"], d["ci"], d["theta"] = ram, ci, theta
    for k, v in zip(["mean_size_rel", "size_95_rel", "pct_block", "size_gini"], out):
        d[k] = np.asarray(v)
    save("trade_size", d)


# ------------------------------------------------------------------ tick-level features
def gen_ticklevel():
    d = {}
    ts, px, am, sd = orc.synth(42, 0, 6_000)
    d["synth"] = np.array([42, 0, 6_000, orc.DENSE_GAP_MOD], dtype=np.int64)
    for w in (1e-6, 0.5, 5.0, 60.0):
        for lg in (False, True):
            d[f"ret_{w}_{int(lg)}"] = rfutils.comp_lagged_returns(ts, px, w, lg)
    # small-timestamp stream (float64 keys exact) incl. a zero price
    ts_s = np.cumsum(np.random.default_rng(1).integers(1, 3_000_000_000, 400)).astype(np.int64)
    px_s = 100.0 + np.cumsum(np.random.default_rng(2).normal(0, 0.1, 400))
    px_s[57] = 0.0
    d["small_ts"], d["small_px"] = ts_s, px_s
    d["small_ret_2.0_0"] = rfutils.comp_lagged_returns(ts_s, px_s, 2.0, False)
    r = rfutils.comp_lagged_returns(ts, px, 5.0, True)
    for hl in (1.0, 30.0, 600.0):
        d[f"ewmst_{hl}"] = rvol.ewmst(ts, r, hl)
        d[f"ewmst0_{hl}"] = rvol.ewmst_mean0(ts, r, hl)
    rn = r.copy()
    rn[1000:1010] = np.nan
    d["ewmst_nan_30.0"] = rvol.ewmst(ts, rn, 30.0)
    for span in (2, 20, 500):
        d[f"ewms_{span}"] = rvol.ewms(rn, span)
    for win, smp in ((2, True), (50, True), (50, False)):
        d[f"rv_{win}_{int(smp)}"] = rvol.realized_vol(rn, win, smp)
    save("ticklevel", d)


def gen_tick_size():
    d = {}
    _, px, _, _ = orc.synth(42, 0, 12_000)
    d["synth_tick"] = np.float64(rutils.comp_price_tick_size(px))
    d["synth"] = np.array([42, 0, 12_000, orc.DENSE_GAP_MOD], dtype=np.int64)
    for i, p in enumerate([np.array([100.0, 100.5, 101.0, 102.5]), np.array([5.0, 5.0, 5.0]),
                           np.array([0.00012, 0.00015, 0.00021, 0.00012]),
                           np.array([27000.1, 27000.3, 26999.9, 27001.7])]):
        d[f"c{i}__px"] = p
        d[f"c{i}__tick"] = np.float64(rutils.comp_price_tick_size(p))
    save("tick_size", d)


def split_trade_stream(seed, n, tick=0.01, second_resolution=False):
    """Raw (un-merged) trades: bursts sharing a timestamp, partly at one price / maker flag."""
    rng = np.random.default_rng(seed)
    burst = rng.geometric(0.45, n)                       # trades per timestamp
    ts = np.repeat(1_700_000_000_000_000_000 + np.cumsum(rng.integers(1, 50_000_000, n)), burst)[:n]
    if second_resolution:
        ts = ts // 1_000_000_000 * 1_000_000_000
    step = rng.choice([-1, 0, 0, 0, 1], n)               # mostly unchanged price inside a burst
    px = (2_700_000 + np.cumsum(step)) * tick
    am = ((1 + rng.integers(0, 4096, n)) * 2.0 ** -10).astype(np.float32)
    am[rng.random(n) < 0.3] = np.float32(0.1)            # inexact float32 sums: the += order matters
    ibm = rng.random(n) < 0.5
    ibm[1:][rng.random(n - 1) < 0.6] = False             # correlate flags so that runs exist
    return ts.astype(np.int64), px, am, ibm


def gen_preprocess():
    d = {}
    for name, seed, n, sec in (("a", 1, 5000, False), ("b", 2, 5000, True), ("c", 3, 300, False)):
        ts, px, am, ibm = split_trade_stream(seed, n, second_resolution=sec)
        for k, v in (("ts", ts), ("px", px), ("am", am), ("ibm", ibm)):
            d[f"{name}__{k}"] = v
        for tag, flag in (("side", ibm), ("noside", None)):
            mts, mpx, mam, msd = rutils.merge_split_trades(ts, px, am, flag)
            d[f"{name}__{tag}_ts"], d[f"{name}__{tag}_px"], d[f"{name}__{tag}_am"] = mts, mpx, mam
            d[f"{name}__{tag}_sd"] = np.asarray(msd, dtype=np.int8)
        d[f"{name}__tickrule"] = rutils.comp_trade_side_vector(px)
    # prices closer than the 1e-8 merge tolerance / the 1e-12 tick-rule epsilon, compared against the HEAD
    px = np.array([1.0, 1.0 + 6e-9, 1.0 + 1.2e-8, 1.0 + 1.8e-8, 1.0 + 1.8e-8, 2.0, 2.0 + 5e-13, 2.0 + 2e-12, 2.0])
    ts = np.full(len(px), 1_700_000_000_000_000_000, dtype=np.int64)
    am = np.full(len(px), 0.1, dtype=np.float32)
    ibm = np.zeros(len(px), dtype=bool)
    mts, mpx, mam, msd = rutils.merge_split_trades(ts, px, am, ibm)
    d["eps__px"], d["eps__ts"], d["eps__am"], d["eps__ibm"] = px, ts, am, ibm
    d["eps__side_ts"], d["eps__side_px"], d["eps__side_am"], d["eps__side_sd"] = mts, mpx, mam, np.asarray(msd, np.int8)
    d["eps__tickrule"] = rutils.comp_trade_side_vector(px)
    save("preprocess", d)


def gen_cusum():
    """_cusum_bar_indexer (logic.py:152-221): EWM sigma with leading / interior NaNs, constant sigma, same-timestamp
    print blocks (a close cannot happen inside one), floor-dominated thresholds."""
    d = {}
    ts, px, am, sd = orc.synth(42, 0, 20_000)
    rng = np.random.default_rng(7)
    px = 100.0 * np.exp(np.cumsum(rng.normal(0, 2e-4, len(ts))))          # returns large enough to trigger closes
    tsb = ts.copy()
    blk = rng.random(len(ts)) < 0.25
    tsb[1:][blk[1:]] = 0
    tsb = np.maximum.accumulate(tsb)                                       # 25 % of the ticks repeat the timestamp
    r = rfutils.comp_lagged_returns(tsb, px, 5.0, True)
    sig = rvol.ewmst(tsb, r, 60.0)
    sig[3000:3040] = np.nan                                                # interior NaNs are forward-filled
    cases = {"ewm": (tsb, px, sig, 5e-4, 2.0), "ewm_lowfloor": (tsb, px, sig, 1e-6, 1.5),
             "const": (ts, px, np.full(len(ts), 1e-3), 5e-4, 2.0),
             "allnan": (ts[:500], px[:500], np.full(500, np.nan), 5e-4, 2.0),
             "floor": (ts, px, np.full(len(ts), 1e-9), 2e-3, 2.0)}
    for name, (t_, p_, s_, fl, mult) in cases.items():
        s_in = s_.copy()
        idx = np.array(rlogic._cusum_bar_indexer(t_, p_, s_in, fl, mult), dtype=np.int64)
        d[f"{name}__ts"], d[f"{name}__px"], d[f"{name}__sigma"] = t_, p_, s_
        d[f"{name}__params"] = np.array([fl, mult])
        d[f"{name}__idx"] = idx
        d[f"{name}__sigma_filled"] = s_in
    save("cusum", d)


def gen_volume_profile():
    """volume_profile_rolling (feature/core/volume.py:403-456) on footprints of the synthetic stream.  Amounts are
    multiples of 2^-4 so that every float32 sum on the way is exact (typed and pure-Python semantics coincide)."""
    from finmlkit.feature.core import volume as rvolume
    from numba.typed import List as NList
    d = {}
    n = 60_000
    ts, px, _, sd = orc.synth(42, 0, n)
    rng = np.random.default_rng(5)
    am = (rng.integers(1, 65, n) * 2.0 ** -4).astype(np.float32)
    for name, interval, window, n_bins in (("m1_w30", 60.0, 1800.0, 27), ("m1_w5_nobins", 60.0, 300.0, None),
                                           ("s10_w120_b5", 10.0, 120.0, 5), ("m1_w30_b200", 60.0, 1800.0, 200)):
        clock, ci = orc._time_bar_indexer(ts, interval)
        o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
        off, flat, _ = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
        nb = len(ci) - 1
        split = lambda a: NList([a[off[i]:off[i + 1]] for i in range(nb)])
        bar_ts = clock[1:]
        poc, hva, lva, pct = rvolume.volume_profile_rolling(bar_ts, o[1], o[2], split(flat["price_levels"]),
                                                            split(flat["buy_volumes"]), split(flat["sell_volumes"]),
                                                            window, n_bins, 0.01, 68.34)
        d[f"{name}__params"] = np.array([interval, window, -1 if n_bins is None else n_bins, 68.34])
        d[f"{name}__poc"], d[f"{name}__hva"], d[f"{name}__lva"], d[f"{name}__pct"] = poc, hva, lva, pct
    d["synth"] = np.array([42, 0, n, orc.DENSE_GAP_MOD], dtype=np.int64)
    d["amount"] = am
    save("volume_profile", d)


def gen_tradesdata():
    """TradesData(preprocess=True) end to end (data_model.py:236-246): raw exchange-style rows in millisecond
    timestamps, shuffled, with duplicated ids and an id gap longer than a minute."""
    from finmlkit.bar.data_model import TradesData
    d = {}
    for name, seed, with_maker, proc_res in (("mk", 11, True, None), ("tr", 12, False, "ms"), ("sec", 13, True, "s")):
        ts_ns, px, am, ibm = split_trade_stream(seed, 4000)
        ts_ms = ts_ns // 1_000_000
        ids = np.arange(len(ts_ms), dtype=np.int64) + 1000
        ids[2500:] += 40                                   # 40 missing ids ...
        ts_ms[2500:] += 120_000                            # ... across a two-minute hole
        rng = np.random.default_rng(seed)
        perm = rng.permutation(len(ids))
        dup = rng.choice(len(ids), 25, replace=False)      # duplicated rows
        order = np.concatenate([perm, dup])
        raw = {"ts": ts_ms[order], "px": px[order], "qty": am[order].astype(np.float64), "id": ids[order]}
        maker = ibm[order] if with_maker else None
        t = TradesData(raw["ts"].copy(), raw["px"].copy(), raw["qty"].copy(), raw["id"].copy(),
                       is_buyer_maker=None if maker is None else maker.copy(), preprocess=True, proc_res=proc_res)
        for k, v in raw.items():
            d[f"{name}__raw_{k}"] = v
        if maker is not None:
            d[f"{name}__raw_maker"] = maker
        d[f"{name}__proc_res"] = np.array(proc_res or "")
        for col in ("timestamp", "price", "amount", "side"):
            d[f"{name}__out_{col}"] = t.data[col].values
        d[f"{name}__data_ok"] = np.array(bool(t.data_ok))
        d[f"{name}__missing_pct"] = np.float64(t.missing_pct)
        d[f"{name}__n_disc"] = np.int64(len(t.discontinuities))
        d[f"{name}__unit"] = np.array(t.orig_timestamp_unit)
    save("tradesdata", d)


if __name__ == "__main__":
    gen_time_indexer()
    gen_threshold_indexers()
    gen_reducers()
    gen_footprint_features()
    gen_trade_size()
    gen_ticklevel()
    gen_tick_size()
    gen_preprocess()
    gen_tradesdata()
    gen_cusum()
    gen_volume_profile()
