"""Per-bar reducers: drop-in for the module-level functions of finmlkit/bar/base.py:306-850.

Same names, positional arguments, output tuples, dtypes and exceptions as the reference; the
arithmetic runs in the HIP kernels of csrc/ through the host-pointer flavour of the C ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
from numpy.typing import NDArray

from .. import _ffi
from .._ffi import c_i64, ptr


def comp_bar_ohlcv(prices: NDArray[np.float64], volumes: NDArray, bar_close_indices: NDArray[np.int64]):
    """Reference: finmlkit/bar/base.py:306-407.

    Returns (open, high, low, close, volume[f32], vwap, trades[i64], median_trade_size)."""
    if len(prices) != len(volumes):
        raise ValueError("Prices and volumes arrays must have the same length.")
    if len(bar_close_indices) < 2:
        raise ValueError("Bar close indices must contain at least two elements.")
    ctx = _ffi.default_context()
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f64 = _ffi.amount_array(volumes)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    nb = len(ci) - 1
    o, h, l, c, vwap, med = (np.empty(nb, np.float64) for _ in range(6))
    vol = np.empty(nb, np.float32)
    tr = np.empty(nb, np.int64)
    ctx.call("fmk_comp_bar_ohlcv", ptr(p), ptr(v), C.c_int(f64), c_i64(len(p)), ptr(ci), c_i64(len(ci)),
             ptr(o), ptr(h), ptr(l), ptr(c), ptr(vol), ptr(vwap), ptr(tr), ptr(med))
    return o, h, l, c, vol, vwap, tr, med


def comp_bar_directional_features(prices: NDArray[np.float64], volumes: NDArray,
                                  bar_close_indices: NDArray[np.int64], trade_sides: NDArray[np.int8]):
    """Reference: finmlkit/bar/base.py:409-546.  Returns the same 14-tuple (dtypes included).

    Raises ZeroDivisionError, like the reference, when a bar has no signed tick (base.py:536)."""
    ctx = _ffi.default_context()
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f64 = _ffi.amount_array(volumes)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    sd = np.ascontiguousarray(trade_sides, dtype=np.int8)
    if len(ci) < 2:
        raise ValueError("Bar close indices must contain at least two elements.")
    nb = len(ci) - 1
    outs = {k: np.empty(nb, dt) for k, dt in _ffi.DIRECTIONAL_FIELDS}
    st = _ffi.DirectionalOut(**{k: a.ctypes.data for k, a in outs.items()})
    ctx.call("fmk_comp_bar_directional", ptr(p), ptr(v), C.c_int(f64), c_i64(len(p)), ptr(ci), c_i64(len(ci)),
             ptr(sd), C.byref(st))
    return tuple(outs[k] for k, _ in _ffi.DIRECTIONAL_FIELDS)


def comp_bar_footprints_csr(prices, amounts, bar_close_indices, trade_sides, price_tick_size, bar_lows,
                            bar_highs, imbalance_factor):
    """CSR form of comp_bar_footprints: (level_offsets[B+1], flat per-level dict, per-bar dict)."""
    ctx = _ffi.default_context()
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f64 = _ffi.amount_array(amounts)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    sd = np.ascontiguousarray(trade_sides, dtype=np.int8)
    lo = np.ascontiguousarray(bar_lows, dtype=np.float64)
    hi = np.ascontiguousarray(bar_highs, dtype=np.float64)
    if len(ci) < 2:
        raise ValueError("Bar close indices must contain at least two elements.")
    nb = len(ci) - 1
    off = np.empty(nb + 1, np.int64)
    args = (ptr(p), ptr(v), C.c_int(f64), c_i64(len(p)), ptr(ci), c_i64(len(ci)), ptr(sd),
            _ffi.c_f64(price_tick_size), ptr(lo), ptr(hi), _ffi.c_f64(imbalance_factor), ptr(off))
    ctx.call("fmk_comp_bar_footprints", *args, None)
    tot = int(off[-1])
    flat = {k: np.empty(tot, dt) for k, dt in _ffi.FOOTPRINT_FLAT_FIELDS}
    bar = {k: np.empty(nb, dt) for k, dt in _ffi.FOOTPRINT_BAR_FIELDS}
    st = _ffi.FootprintOut(**{k: a.ctypes.data for k, a in {**flat, **bar}.items()})
    ctx.call("fmk_comp_bar_footprints", *args, C.byref(st))
    return off, flat, bar


def comp_bar_footprints(prices, amounts, bar_close_indices, trade_sides, price_tick_size, bar_lows, bar_highs,
                        imbalance_factor):
    """Reference: finmlkit/bar/base.py:615-752.  Same 13-tuple: seven lists of per-bar arrays
    (views into the CSR buffers) followed by six per-bar arrays."""
    off, flat, bar = comp_bar_footprints_csr(prices, amounts, bar_close_indices, trade_sides, price_tick_size,
                                             bar_lows, bar_highs, imbalance_factor)
    nb = len(off) - 1

    def split(a):
        return [a[off[i]:off[i + 1]] for i in range(nb)]
    return (split(flat["price_levels"]), split(flat["buy_volumes"]), split(flat["sell_volumes"]),
            split(flat["buy_ticks"]), split(flat["sell_ticks"]),
            split(flat["buy_imbalances"].view(np.bool_)), split(flat["sell_imbalances"].view(np.bool_)),
            bar["buy_imbalances_sum"], bar["sell_imbalances_sum"], bar["cot_price_levels"],
            bar["imb_max_run_signed"], bar["vp_skew"], bar["vp_gini"])
