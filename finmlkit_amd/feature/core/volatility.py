"""Drop-in for the tick-level estimators of finmlkit/feature/core/volatility.py on the MI355X.

The estimators that run on the raw tick frame are on the hot path: `ewmst`,
`ewmst_mean0`, `ewms` and `realized_vol`.  The bar-level indicators of that module (Bollinger, Parkinson,
ATR, variance ratio) are out of scope.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
from numpy.typing import NDArray

from ... import _ffi
from ..._ffi import c_f64, c_i64, ptr


def _ewmst(timestamps, y, half_life, sigma_floor, mean0):
    ctx = _ffi.default_context()
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    yy = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(len(yy), np.float64)
    ctx.call("fmk_ewmst", ptr(ts), ptr(yy), c_i64(len(yy)), c_f64(half_life), c_f64(sigma_floor),
             C.c_int(int(mean0)), ptr(out))
    return out


def ewmst(timestamps: NDArray[np.int64], y: NDArray[np.float64], half_life: float,
          sigma_floor: float = 1e-12) -> NDArray[np.float64]:
    """Reference: finmlkit/feature/core/volatility.py:139-219 (unbiased time-decay EWMA std)."""
    return _ewmst(timestamps, y, half_life, sigma_floor, False)


def ewmst_mean0(timestamps: NDArray[np.int64], y: NDArray[np.float64], half_life: float,
                sigma_floor: float = 1e-12) -> NDArray[np.float64]:
    """Reference: finmlkit/feature/core/volatility.py:72-136 (zero-mean variant)."""
    return _ewmst(timestamps, y, half_life, sigma_floor, True)


def ewms(y: NDArray[np.float64], span: int) -> NDArray[np.float64]:
    """Reference: finmlkit/feature/core/volatility.py:9-69 (fixed-alpha EW std, alpha = 2/(span+1))."""
    ctx = _ffi.default_context()
    yy = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(len(yy), np.float64)
    ctx.call("fmk_ewms", ptr(yy), c_i64(len(yy)), c_i64(int(span)), ptr(out))
    return out


def realized_vol(r: NDArray[np.float64], window: int, is_sample: bool) -> NDArray[np.float64]:
    """Reference: finmlkit/feature/core/volatility.py:256-286 (rolling sqrt(nansum(r^2) / (valid - is_sample)))."""
    ctx = _ffi.default_context()
    rr = np.ascontiguousarray(r, dtype=np.float64)
    if int(window) == 0:                           # every window is empty: NaN everywhere (nothing to compute)
        return np.full(len(rr), np.nan)
    out = np.empty(len(rr), np.float64)
    ctx.call("fmk_realized_vol", ptr(rr), c_i64(len(rr)), c_i64(int(window)), C.c_int(bool(is_sample)), ptr(out))
    return out
