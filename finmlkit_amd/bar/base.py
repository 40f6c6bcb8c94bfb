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
