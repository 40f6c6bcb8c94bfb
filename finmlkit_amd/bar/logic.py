"""Bar close-index builders: drop-in for finmlkit/bar/logic.py, computed on the MI355X.

Same names, arguments, return values and error behaviour as the reference functions; each
call uploads its NumPy inputs, runs the HIP kernels of csrc/fmk_indexers.hip /
csrc/fmk_threshold.hip through the C ABI and downloads the result.  For device-resident
pipelines use finmlkit_amd.engine.DeviceTrades instead (no PCIe traffic per call).
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
from numpy.typing import NDArray

from .. import _ffi
from .._ffi import DeviceArray, c_f64, c_i64, ptr


def _time_bar_indexer(timestamps: NDArray[np.int64], interval_seconds: float
                      ) -> Tuple[NDArray[np.int64], NDArray[np.int64]]:
    """Reference: finmlkit/bar/logic.py:12-51.  Returns (bar_clock, bar_close_indices)."""
    ctx = _ffi.default_context()
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    if interval_seconds < 0:
        # logic.py:33-39 with a negative step: np.arange(start, last + I + 1, I) is empty whenever the stream spans more
        # than |I| (edge sweep: the reference returns two empty arrays); anything else is not supported
        I = interval_seconds * 1e9
        clock = np.arange(float(ts[0]) // I * I, np.ceil(ts[-1] / I) * I + I + 1, I, dtype=np.int64) if len(ts) else None
        if clock is not None and len(clock) == 0:
            return np.empty(0, np.int64), np.empty(0, np.int64)
        raise ValueError("interval_seconds must be positive")
    ne = c_i64()
    ctx.call("fmk_time_bar_indexer", ptr(ts), c_i64(len(ts)), c_f64(interval_seconds), None, None, c_i64(0),
             C.byref(ne))
    clock = np.empty(ne.value, np.int64)
    idx = np.empty(ne.value, np.int64)
    ctx.call("fmk_time_bar_indexer", ptr(ts), c_i64(len(ts)), c_f64(interval_seconds), ptr(clock), ptr(idx),
             c_i64(ne.value), C.byref(ne))
    return clock, idx


def _tick_bar_indexer(timestamps: NDArray[np.int64], threshold: int) -> NDArray[np.int64]:
    """Reference: finmlkit/bar/logic.py:54-84 (returns an int64 array instead of a numba List)."""
    ctx = _ffi.default_context()
    if len(timestamps) == 0:                       # logic.py:54-84: the list starts as [0] and the loop does not run
        return np.zeros(1, np.int64)
    m = c_i64()
    ctx.call("fmk_tick_bar_indexer_dev", c_i64(len(timestamps)), c_i64(int(threshold)), None, c_i64(0),
             C.byref(m))
    out = DeviceArray(ctx, m.value, np.int64)
    ctx.call("fmk_tick_bar_indexer_dev", c_i64(len(timestamps)), c_i64(int(threshold)), out.p, c_i64(m.value),
             C.byref(m))
    return out.to_host()


def _threshold_indexer(fn, cols, n, threshold):
    ctx = _ffi.default_context()
    m, unc = c_i64(), c_i64()
    ctx.call(fn, *cols, c_i64(n), c_f64(threshold), None, c_i64(0), C.byref(m), C.byref(unc))
    out = DeviceArray(ctx, m.value, np.int64)
    ctx.call(fn, *cols, c_i64(n), c_f64(threshold), out.p, c_i64(m.value), C.byref(m), C.byref(unc))
    return out.to_host()


def _volume_bar_indexer(volumes: NDArray, threshold: float) -> NDArray[np.int64]:
    """Reference: finmlkit/bar/logic.py:87-115."""
    ctx = _ffi.default_context()
    v, f64 = _ffi.amount_array(volumes)
    dv = DeviceArray.from_host(ctx, v)
    return _threshold_indexer("fmk_volume_bar_indexer_dev", (dv.p, C.c_int(f64)), len(v), threshold)


def _dollar_bar_indexer(prices: NDArray[np.float64], volumes: NDArray, threshold: float) -> NDArray[np.int64]:
    """Reference: finmlkit/bar/logic.py:118-149."""
    ctx = _ffi.default_context()
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f64 = _ffi.amount_array(volumes)
    dp = DeviceArray.from_host(ctx, p)
    dv = DeviceArray.from_host(ctx, v)
    return _threshold_indexer("fmk_dollar_bar_indexer_dev", (dp.p, dv.p, C.c_int(f64)), len(v), threshold)


def _imbalance_bar_indexer(timestamps, prices, volumes, threshold):
    """Reference stub: finmlkit/bar/logic.py:224-241."""
    raise NotImplementedError("Imbalance bar indexer is not implemented yet.")


def _run_bar_indexer(timestamps, prices, volumes, threshold):
    """Reference stub: finmlkit/bar/logic.py:244-261."""
    raise NotImplementedError("Run bar indexer is not implemented yet.")


def _cusum_bar_indexer(timestamps: NDArray[np.int64], prices: NDArray[np.float64], sigma: NDArray[np.float64],
                       sigma_floor: float, sigma_mult: float) -> NDArray[np.int64]:
    """Symmetric CUSUM filter on log price changes (reference logic.py:152-221): a bar closes when the positive /
    negative cumulative sum reaches +-max(sigma_mult * sigma[i], sigma_floor) at a tick that is not followed by a
    same-timestamp tick.  Like the reference, NaNs of `sigma` are forward-filled IN PLACE from its first valid entry,
    whose index opens the result.  Parallel-in-time fixed point on the device (csrc/fmk_cusum.hip)."""
    import ctypes as C
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    px = np.ascontiguousarray(prices, dtype=np.float64)
    # logic.py:174-175 is the CHAINED comparison len(prices) != len(sigma) != len(timestamps): it raises only when both
    # inequalities hold; otherwise n = len(prices) and longer sigma / timestamps are read up to n (oracle/edge_sweep.py)
    if len(px) != len(sigma) and len(sigma) != len(ts):
        raise ValueError("Prices, timestamps, and sigma arrays must have the same length.")
    if len(px) == 0:
        return np.zeros(1, np.int64)                # the list starts with first_non_nan_idx = 0, the loop does not run
    if len(ts) < len(px) or len(sigma) < len(px):   # the reference indexes past the shorter array here
        raise ValueError("timestamps / sigma shorter than prices")
    ts = np.ascontiguousarray(ts[:len(px)])
    sigma = sigma[:len(px)]                         # a view: the forward fill below still lands in the caller's array
    sg = sigma if (isinstance(sigma, np.ndarray) and sigma.dtype == np.float64 and sigma.flags["C_CONTIGUOUS"]
                   and sigma.flags["WRITEABLE"]) else np.array(sigma, dtype=np.float64)
    n = len(px)
    out = np.empty(n, np.int64)
    m = c_i64()
    _ffi.default_context().call("fmk_cusum_bar_indexer", ptr(ts), ptr(px), ptr(sg), c_i64(n), c_f64(sigma_floor),
                                c_f64(sigma_mult), ptr(out), c_i64(n), C.byref(m))
    if sg is not sigma and isinstance(sigma, np.ndarray) and sigma.flags["WRITEABLE"]:
        sigma[...] = sg                                   # keep the in-place fill visible to the caller
    return out[:m.value].copy()
