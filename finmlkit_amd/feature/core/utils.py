"""Drop-in for finmlkit/feature/core/utils.py::comp_lagged_returns, computed on the MI355X."""
from __future__ import annotations

import ctypes as C

import numpy as np
from numpy.typing import NDArray

from ... import _ffi
from ..._ffi import c_f64, c_i64, ptr


def comp_lagged_returns(timestamps: NDArray[np.int64], close: NDArray[np.float64], return_window_sec: float,
                        is_log: bool) -> NDArray[np.float64]:
    """Reference: finmlkit/feature/core/utils.py:12-64 (float64 key comparison semantics included)."""
    if return_window_sec <= 0:
        raise ValueError("The return window must be greater than zero.")
    ctx = _ffi.default_context()
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    c = np.ascontiguousarray(close, dtype=np.float64)
    out = np.empty(len(c), np.float64)
    ctx.call("fmk_comp_lagged_returns", ptr(ts), ptr(c), c_i64(len(c)), c_f64(return_window_sec),
             C.c_int(bool(is_log)), ptr(out))
    return out
