"""ctypes front-end of the CPU parity oracle (oracle/fmk_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the finmlkit_amd package.  Function
names and argument order mirror the reference functions they restate so the
parity tests read like the reference's own tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libfmk_oracle.so")

OK, E_ARG, E_CAPACITY, E_LEVEL, E_ZERODIV, E_NOMEM = 0, -1, -2, -3, -4, -5


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Returns the .so path."""
    src = os.path.join(_HERE, "fmk_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_tick_bar_indexer.restype = C.c_int64
        _lib.orc_volume_bar_indexer.restype = C.c_int64
        _lib.orc_dollar_bar_indexer.restype = C.c_int64
        _lib.orc_cusum_bar_indexer.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _amt(a) -> Tuple[np.ndarray, int]:
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a, 0
    return np.ascontiguousarray(a, dtype=np.float64), 1


def _i64(x):
    return C.c_int64(int(x))


def _f64(x):
    return C.c_double(float(x))


def _check(rc, allow=()):
    if rc == OK or rc in allow:
        return
    if rc in (E_ARG, E_CAPACITY, E_LEVEL):
        raise ValueError(f"oracle status {rc}")
    if rc == E_ZERODIV:
        raise ZeroDivisionError("division by zero")
    raise RuntimeError(f"oracle status {rc}")


# ---------------------------------------------------------------- synthetic stream
DENSE_GAP_MOD = 100_000_000
SPARSE_GAP_MOD = 500_000_000_000


def synth(seed: int, first: int, n: int, gap_mod: int = DENSE_GAP_MOD):
    ts = np.empty(n, np.int64)
    px = np.empty(n, np.float64)
    am = np.empty(n, np.float32)
    sd = np.empty(n, np.int8)
    _check(lib().orc_synth(C.c_uint64(seed), _i64(first), _i64(n), C.c_uint64(gap_mod),
                           _p(ts), _p(px), _p(am), _p(sd)))
    return ts, px, am, sd


# ---------------------------------------------------------------- indexers
def time_bar_clock(ts_first: int, ts_last: int, interval_seconds: float):
    ne, e0, d = C.c_int64(), C.c_int64(), C.c_int64()
    _check(lib().orc_time_bar_clock(_i64(ts_first), _i64(ts_last), _f64(interval_seconds),
                                    C.byref(ne), C.byref(e0), C.byref(d)))
    return ne.value, e0.value, d.value


def _time_bar_indexer(timestamps, interval_seconds):
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    if interval_seconds < 0:
        # logic.py:33-39 with a negative step: np.arange(start, last + I + 1, I) is empty whenever the stream spans more
        # than |I| (edge sweep: the reference returns two empty arrays); anything else is not supported
        I = interval_seconds * 1e9
        clock = np.arange(float(ts[0]) // I * I, np.ceil(ts[-1] / I) * I + I + 1, I, dtype=np.int64) if len(ts) else None
        if clock is not None and len(clock) == 0:
            return np.empty(0, np.int64), np.empty(0, np.int64)
        raise ValueError("interval_seconds must be positive")
    ne = C.c_int64()
    _check(lib().orc_time_bar_indexer(_p(ts), _i64(len(ts)), _f64(interval_seconds), None, None,
                                      _i64(0), C.byref(ne)))
    clock = np.empty(ne.value, np.int64)
    idx = np.empty(ne.value, np.int64)
    _check(lib().orc_time_bar_indexer(_p(ts), _i64(len(ts)), _f64(interval_seconds), _p(clock),
                                      _p(idx), _i64(ne.value), C.byref(ne)))
    return clock, idx


def _two_phase(fn, *args):
    m = fn(*args, None, _i64(0))
    if m < 0:
        _check(int(m))
    out = np.empty(m, np.int64)
    m2 = fn(*args, _p(out), _i64(m))
    assert m2 == m
    return out


def _tick_bar_indexer(timestamps, threshold):
    if len(timestamps) == 0:                       # logic.py:54-84: the list starts as [0] and the loop does not run
        return np.zeros(1, np.int64)
    return _two_phase(lib().orc_tick_bar_indexer, _i64(len(timestamps)), _i64(threshold))


def _volume_bar_indexer(volumes, threshold):
    v, f = _amt(volumes)
    return _two_phase(lib().orc_volume_bar_indexer, _p(v), C.c_int(f), _i64(len(v)), _f64(threshold))


def _dollar_bar_indexer(prices, volumes, threshold):
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f = _amt(volumes)
    return _two_phase(lib().orc_dollar_bar_indexer, _p(p), _p(v), C.c_int(f), _i64(len(v)),
                      _f64(threshold))


def _cusum_bar_indexer(timestamps, prices, sigma, sigma_floor, sigma_mult, return_sigma=False):
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    p = np.ascontiguousarray(prices, dtype=np.float64)
    # logic.py:174-175 is the CHAINED comparison len(prices) != len(sigma) != len(timestamps): it raises only when both
    # inequalities hold; otherwise n = len(prices) and longer sigma / timestamps are read up to n (edge sweep)
    if len(p) != len(sigma) and len(sigma) != len(ts):
        raise ValueError("Prices, timestamps, and sigma arrays must have the same length.")
    n = len(p)
    if n == 0:
        return (np.zeros(1, np.int64), np.array(sigma, dtype=np.float64)) if return_sigma else np.zeros(1, np.int64)
    if len(ts) < n or len(sigma) < n:              # the reference indexes past the shorter array here
        raise ValueError("timestamps / sigma shorter than prices")
    ts = ts[:n]
    sigma = np.asarray(sigma)[:n]
    s1 = np.array(sigma, dtype=np.float64)
    m = lib().orc_cusum_bar_indexer(_p(ts), _p(p), _p(s1), _i64(len(p)), _f64(sigma_floor),
                                    _f64(sigma_mult), None, _i64(0))
    out = np.empty(m, np.int64)
    s2 = np.array(sigma, dtype=np.float64)
    lib().orc_cusum_bar_indexer(_p(ts), _p(p), _p(s2), _i64(len(p)), _f64(sigma_floor),
                                _f64(sigma_mult), _p(out), _i64(m))
    return (out, s2) if return_sigma else out          # s2: sigma forward-filled like the reference does in place


# ---------------------------------------------------------------- reducers
def comp_bar_ohlcv(prices, volumes, bar_close_indices, want_median: bool = True):
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f = _amt(volumes)
    if len(p) != len(v):
        raise ValueError("Prices and volumes arrays must have the same length.")
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    if len(ci) < 2:
        raise ValueError("Bar close indices must contain at least two elements.")
    nb = len(ci) - 1
    o, h, l, c = (np.zeros(nb, np.float64) for _ in range(4))
    vol = np.zeros(nb, np.float32)
    vwap = np.zeros(nb, np.float64)
    tr = np.zeros(nb, np.int64)
    med = np.zeros(nb, np.float64)
    _check(lib().orc_comp_bar_ohlcv(_p(p), _p(v), C.c_int(f), _i64(len(p)), _p(ci), _i64(len(ci)),
                                    _p(o), _p(h), _p(l), _p(c), _p(vol), _p(vwap), _p(tr),
                                    _p(med) if want_median else None))
    return o, h, l, c, vol, vwap, tr, med


def comp_bar_directional_features(prices, volumes, bar_close_indices, trade_sides,
                                  raise_on_zero_div: bool = True):
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f = _amt(volumes)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    sd = np.ascontiguousarray(trade_sides, dtype=np.int8)
    nb = len(ci) - 1
    i64 = lambda: np.zeros(nb, np.int64)
    f32 = lambda: np.zeros(nb, np.float32)
    outs = (i64(), i64(), f32(), f32(), f32(), f32(), f32(), f32(), i64(), i64(), f32(), f32(),
            f32(), f32())
    rc = lib().orc_comp_bar_directional(_p(p), _p(v), C.c_int(f), _i64(len(p)), _p(ci),
                                        _i64(len(ci)), _p(sd), *[_p(a) for a in outs])
    _check(rc, allow=() if raise_on_zero_div else (E_ZERODIV,))
    return outs


def comp_bar_trade_size_features(amounts, theta, bar_close_indices, theta_mult):
    v, f = _amt(amounts)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    if len(th) != len(ci) - 1:
        raise ValueError("Theta should match the the number of bars (len(bar_close_indices) - 1).")
    nb = len(ci) - 1
    outs = tuple(np.zeros(nb, np.float32) for _ in range(4))
    _check(lib().orc_comp_bar_trade_size(_p(v), C.c_int(f), _i64(len(v)), _p(th), _p(ci),
                                         _i64(len(ci)), _f64(theta_mult), *[_p(a) for a in outs]))
    return outs


def comp_footprint_features(price_levels, buy_volumes, sell_volumes, imbalance_multiplier):
    lv = np.ascontiguousarray(price_levels, dtype=np.int32)
    b = np.ascontiguousarray(buy_volumes, dtype=np.float32)
    s = np.ascontiguousarray(sell_volumes, dtype=np.float32)
    L = len(lv)
    bi = np.zeros(L, np.uint8)
    si = np.zeros(L, np.uint8)
    run, cot = C.c_int32(), C.c_int32()
    sk, gi = C.c_double(), C.c_double()
    _check(lib().orc_comp_footprint_features(_p(lv), _p(b), _p(s), _i64(L), _f64(imbalance_multiplier),
                                             _p(bi), _p(si), C.byref(run), C.byref(cot),
                                             C.byref(sk), C.byref(gi)))
    return bi.astype(bool), si.astype(bool), run.value, cot.value, sk.value, gi.value


def comp_bar_footprints_csr(prices, amounts, bar_close_indices, trade_sides, price_tick_size,
                            bar_lows, bar_highs, imbalance_factor):
    """CSR form of comp_bar_footprints: returns (level_offsets, flat dict, per-bar dict)."""
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f = _amt(amounts)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    sd = np.ascontiguousarray(trade_sides, dtype=np.int8)
    lo = np.ascontiguousarray(bar_lows, dtype=np.float64)
    hi = np.ascontiguousarray(bar_highs, dtype=np.float64)
    nb = len(ci) - 1
    off = np.zeros(nb + 1, np.int64)
    args = (_p(p), _p(v), C.c_int(f), _i64(len(p)), _p(ci), _i64(len(ci)), _p(sd),
            _f64(price_tick_size), _p(lo), _p(hi), _f64(imbalance_factor), _p(off))
    _check(lib().orc_comp_bar_footprints(*args, *([None] * 13)))
    tot = int(off[-1])
    flat = dict(price_levels=np.zeros(tot, np.int32), buy_volumes=np.zeros(tot, np.float32),
                sell_volumes=np.zeros(tot, np.float32), buy_ticks=np.zeros(tot, np.int32),
                sell_ticks=np.zeros(tot, np.int32), buy_imbalances=np.zeros(tot, np.uint8),
                sell_imbalances=np.zeros(tot, np.uint8))
    bar = dict(buy_imbalances_sum=np.zeros(nb, np.uint16), sell_imbalances_sum=np.zeros(nb, np.uint16),
               cot_price_levels=np.zeros(nb, np.int32), imb_max_run_signed=np.zeros(nb, np.int16),
               vp_skew=np.zeros(nb, np.float64), vp_gini=np.zeros(nb, np.float64))
    _check(lib().orc_comp_bar_footprints(*args, *[_p(a) for a in flat.values()],
                                         *[_p(a) for a in bar.values()]))
    return off, flat, bar


def comp_bar_footprints(prices, amounts, bar_close_indices, trade_sides, price_tick_size,
                        bar_lows, bar_highs, imbalance_factor):
    """Reference-shaped 13-tuple (lists of per-bar arrays + per-bar arrays)."""
    off, flat, bar = comp_bar_footprints_csr(prices, amounts, bar_close_indices, trade_sides,
                                             price_tick_size, bar_lows, bar_highs, imbalance_factor)
    nb = len(off) - 1

    def split(a, dt=None):
        return [a[off[i]:off[i + 1]].astype(dt) if dt else a[off[i]:off[i + 1]].copy()
                for i in range(nb)]
    return (split(flat["price_levels"]), split(flat["buy_volumes"]), split(flat["sell_volumes"]),
            split(flat["buy_ticks"]), split(flat["sell_ticks"]),
            split(flat["buy_imbalances"], bool), split(flat["sell_imbalances"], bool),
            bar["buy_imbalances_sum"], bar["sell_imbalances_sum"], bar["cot_price_levels"],
            bar["imb_max_run_signed"], bar["vp_skew"], bar["vp_gini"])


# ---------------------------------------------------------------- tick-level features
def comp_lagged_returns(timestamps, close, return_window_sec, is_log):
    if return_window_sec <= 0:
        raise ValueError("The return window must be greater than zero.")
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    c = np.ascontiguousarray(close, dtype=np.float64)
    out = np.empty(len(c), np.float64)
    _check(lib().orc_comp_lagged_returns(_p(ts), _p(c), _i64(len(c)), _f64(return_window_sec),
                                         C.c_int(bool(is_log)), _p(out)))
    return out


def ewmst(timestamps, y, half_life, sigma_floor=1e-12):
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    yy = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(len(yy), np.float64)
    _check(lib().orc_ewmst(_p(ts), _p(yy), _i64(len(yy)), _f64(half_life), _f64(sigma_floor), _p(out)))
    return out


def ewmst_mean0(timestamps, y, half_life, sigma_floor=1e-12):
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    yy = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(len(yy), np.float64)
    _check(lib().orc_ewmst_mean0(_p(ts), _p(yy), _i64(len(yy)), _f64(half_life), _f64(sigma_floor),
                                 _p(out)))
    return out


def ewms(y, span):
    yy = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(len(yy), np.float64)
    _check(lib().orc_ewms(_p(yy), _i64(len(yy)), _i64(span), _p(out)))
    return out


def realized_vol(r, window, is_sample):
    rr = np.ascontiguousarray(r, dtype=np.float64)
    if window == 0:                                # volatility.py:256-286: every window is empty -> NaN everywhere
        return np.full(len(rr), np.nan)
    out = np.empty(len(rr), np.float64)
    _check(lib().orc_realized_vol(_p(rr), _i64(len(rr)), _i64(window), C.c_int(bool(is_sample)), _p(out)))
    return out


def comp_price_tick_size(prices):
    p = np.ascontiguousarray(prices, dtype=np.float64)
    if len(p) == 0:
        raise ValueError("Empty prices array")
    out = C.c_double()
    _check(lib().orc_comp_price_tick_size(_p(p), _i64(len(p)), C.byref(out)))
    return out.value


def merge_split_trades(timestamps, prices, amounts, is_buyer_maker):
    """bar/utils.py:263-329 -> (ts, price, amount float32, side int8 | empty)."""
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    px = np.ascontiguousarray(prices, dtype=np.float64)
    am = np.ascontiguousarray(amounts, dtype=np.float32)
    n = len(ts)
    ibm = None if is_buyer_maker is None else np.ascontiguousarray(is_buyer_maker, dtype=np.uint8)
    o_ts, o_px, o_am = np.empty(n, np.int64), np.empty(n, np.float64), np.empty(n, np.float32)
    o_sd = np.empty(n, np.int8)
    fn = lib().orc_merge_split_trades
    fn.restype = C.c_int64
    m = fn(_p(ts), _p(px), _p(am), _p(ibm), _i64(n), _p(o_ts), _p(o_px), _p(o_am), _p(o_sd))
    if m < 0:
        _check(int(m))
    return o_ts[:m], o_px[:m], o_am[:m], (o_sd[:m] if ibm is not None else np.empty(0, dtype=np.int8))


def comp_trade_side_vector(prices):
    """bar/utils.py:26-46 (tick rule)."""
    px = np.ascontiguousarray(prices, dtype=np.float64)
    out = np.empty(len(px), np.int8)
    _check(lib().orc_comp_trade_side_vector(_p(px), _i64(len(px)), _p(out)))
    return out


def volume_profile_rolling(ts, highs, lows, level_offsets, price_levels, buy_volumes, sell_volumes, window_size_sec,
                           n_bins=None, price_tick=None, va_pct=68.34):
    """feature/core/volume.py:403-456 on CSR footprints -> (poc, hva, lva int32, pct_above_poc float32)."""
    t = np.ascontiguousarray(ts, dtype=np.int64)
    hi = np.ascontiguousarray(highs, dtype=np.float64)
    lo = np.ascontiguousarray(lows, dtype=np.float64)
    off = np.ascontiguousarray(level_offsets, dtype=np.int64)
    pl = np.ascontiguousarray(price_levels, dtype=np.int32)
    bv = np.ascontiguousarray(buy_volumes, dtype=np.float32)
    sv = np.ascontiguousarray(sell_volumes, dtype=np.float32)
    nb = len(t)
    poc, hva, lva = (np.zeros(nb, np.int32) for _ in range(3))
    pct = np.zeros(nb, np.float32)
    _check(lib().orc_volume_profile_rolling(_p(t), _p(hi), _p(lo), _p(off), _p(pl), _p(bv), _p(sv), _i64(nb),
                                            _i64(int(window_size_sec * 1e9)), _i64(-1 if n_bins is None else n_bins),
                                            _f64(price_tick), _f64(va_pct), _p(poc), _p(hva), _p(lva), _p(pct)))
    return poc, hva, lva, pct


def calc_volume_percentage_above_poc(price_levels, volumes, poc_price):
    """finmlkit/feature/core/volume.py:367-391."""
    pl = np.ascontiguousarray(price_levels, dtype=np.int32)
    v = np.ascontiguousarray(volumes, dtype=np.float32)
    f = lib().orc_calc_volume_percentage_above_poc
    f.restype = C.c_double
    return float(f(_p(pl), _p(v), _i64(len(pl)), C.c_int32(int(poc_price))))


def resample_bars(seg, open_, high, low, close, volume, trades, vwap, median):
    """TimeBarReader._resample (finmlkit/bar/io.py:890-950) on contiguous row groups [seg[g], seg[g+1])
    -> (open, high, low, close, volume, trades, vwap f32, median f32, valid)."""
    seg = np.ascontiguousarray(seg, dtype=np.int64)
    G = len(seg) - 1
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    o, h, l, c, m = f8(open_), f8(high), f8(low), f8(close), f8(median)
    vol = np.ascontiguousarray(volume, dtype=np.float32 if np.asarray(volume).dtype == np.float32 else np.float64)
    vw = np.ascontiguousarray(vwap, dtype=np.float32 if np.asarray(vwap).dtype == np.float32 else np.float64)
    tr = np.ascontiguousarray(trades, dtype=np.int64)
    out = [np.empty(G, np.float64) for _ in range(4)] + [np.empty(G, vol.dtype), np.empty(G, np.int64),
                                                         np.empty(G, np.float32), np.empty(G, np.float32), np.empty(G, np.uint8)]
    _check(lib().orc_resample_bars(_p(seg), _i64(G), _p(o), _p(h), _p(l), _p(c), _p(vol), C.c_int(vol.dtype == np.float64),
                                   _p(tr), _p(vw), C.c_int(vw.dtype == np.float64), _p(m), *[_p(a) for a in out]))
    return tuple(out)
