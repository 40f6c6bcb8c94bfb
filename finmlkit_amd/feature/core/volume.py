"""Rolling volume profile on the MI355X: drop-in for `VolumePro` / `volume_profile_rolling` of
finmlkit/feature/core/volume.py (:12-131, :403-456).

The reference walks ragged `numba.typed.List`s of per-bar level arrays; here the footprints stay in the
CSR layout the footprint kernel produces (`FootprintData.level_offsets` + `.flat`) and one wave per bar
aggregates its window in LDS (csrc/fmk_volprofile.hip).  Ragged lists are accepted too (concatenated
once on the host).  Numba-typed semantics: float32 level sums in the reference's order, float64 scalars.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd
from numpy.typing import NDArray

from ... import _ffi
from ..._ffi import c_f64, c_i64, ptr


def _to_csr(price_levels: Sequence, buy_volumes: Sequence, sell_volumes: Sequence):
    n = len(price_levels)
    off = np.zeros(n + 1, dtype=np.int64)
    if n:
        np.cumsum([len(a) for a in price_levels], out=off[1:])
    cat = lambda parts, dt: (np.ascontiguousarray(np.concatenate([np.asarray(p) for p in parts]), dtype=dt)
                             if n and off[-1] else np.empty(0, dtype=dt))
    return off, cat(price_levels, np.int32), cat(buy_volumes, np.float32), cat(sell_volumes, np.float32)


def volume_profile_rolling_csr(ts: NDArray[np.int64], highs: NDArray[np.float64], lows: NDArray[np.float64],
                               level_offsets: NDArray[np.int64], price_levels: NDArray[np.int32],
                               buy_volumes: NDArray[np.float32], sell_volumes: NDArray[np.float32],
                               window_size_sec: float, n_bins: Optional[int] = None, price_tick: float = None,
                               va_pct: float = 68.34) -> Tuple[NDArray, NDArray, NDArray, NDArray]:
    """`volume_profile_rolling` on CSR footprints -> (poc, hva, lva int32 in tick units, share above POC float32)."""
    t = np.ascontiguousarray(ts, dtype=np.int64)
    hi = np.ascontiguousarray(highs, dtype=np.float64)
    lo = np.ascontiguousarray(lows, dtype=np.float64)
    off = np.ascontiguousarray(level_offsets, dtype=np.int64)
    nb = len(t)
    if not (len(hi) == len(lo) == nb == len(off) - 1) or nb == 0:
        raise AssertionError("Input arrays should have the same length and be non-empty.")
    if price_tick is None:
        raise ValueError("price_tick is required")
    pl = np.ascontiguousarray(price_levels, dtype=np.int32)
    bv = np.ascontiguousarray(buy_volumes, dtype=np.float32)
    sv = np.ascontiguousarray(sell_volumes, dtype=np.float32)
    window_ns = int(window_size_sec * 1e9)                                   # volume.py:430
    first = int(np.searchsorted(t, t[0] + window_ns))                        # volume.py:432
    poc, hva, lva = (np.zeros(nb, dtype=np.int32) for _ in range(3))
    pct = np.zeros(nb, dtype=np.float32)
    _ffi.default_context().call("fmk_volume_profile_rolling", ptr(t), ptr(hi), ptr(lo), ptr(off), ptr(pl), ptr(bv),
                                ptr(sv), c_i64(nb), c_i64(first), c_i64(window_ns),
                                c_i64(-1 if n_bins is None else int(n_bins)), c_f64(price_tick), c_f64(va_pct),
                                ptr(poc), ptr(hva), ptr(lva), ptr(pct))
    return poc, hva, lva, pct


def volume_profile_rolling(ts, highs, lows, price_levels, buy_volumes, sell_volumes, window_size_sec: float,
                           n_bins: int = None, price_tick: float = None, va_pct: float = 68.34):
    """Reference signature (volume.py:403-408): ragged per-bar lists of level arrays."""
    assert len(ts) == len(highs) == len(lows) == len(price_levels) == len(buy_volumes) == len(sell_volumes) > 0, \
        "Input arrays should have the same length and be non-empty."
    off, pl, bv, sv = _to_csr(price_levels, buy_volumes, sell_volumes)
    return volume_profile_rolling_csr(ts, highs, lows, off, pl, bv, sv, window_size_sec, n_bins, price_tick, va_pct)


def aggregate_footprint(ts, highs, lows, price_levels, buy_volumes, sell_volumes, start_ts: int, end_ts: int,
                        price_tick: float, level_offsets=None):
    """Reference signature (volume.py:134-140): the footprints of the bars in [start_ts, end_ts] summed on one dense level grid ->
    (complete_price_levels int32, aligned_buy_volumes float32, aligned_sell_volumes float32).  `price_levels` / volumes: the
    reference's ragged per-bar lists, or -- with `level_offsets` -- the flat CSR arrays of `FootprintData`."""
    import ctypes as C
    t = np.ascontiguousarray(ts, dtype=np.int64)
    hi = np.ascontiguousarray(highs, dtype=np.float64)
    lo = np.ascontiguousarray(lows, dtype=np.float64)
    if level_offsets is None:
        off, pl, bv, sv = _to_csr(price_levels, buy_volumes, sell_volumes)
    else:
        off = np.ascontiguousarray(level_offsets, dtype=np.int64)
        pl = np.ascontiguousarray(price_levels, dtype=np.int32)
        bv = np.ascontiguousarray(buy_volumes, dtype=np.float32)
        sv = np.ascontiguousarray(sell_volumes, dtype=np.float32)
    nb = len(t)
    if not (len(hi) == len(lo) == nb == len(off) - 1) or nb == 0:
        raise AssertionError("Input arrays should have the same length and be non-empty.")
    ctx = _ffi.default_context()
    minl, nlev = C.c_int32(), c_i64()
    args = (ptr(t), ptr(hi), ptr(lo), ptr(off), ptr(pl), ptr(bv), ptr(sv), c_i64(nb), c_i64(int(start_ts)), c_i64(int(end_ts)),
            c_f64(price_tick), C.byref(minl), C.byref(nlev))
    ctx.call("fmk_aggregate_footprint", *args, None, None, c_i64(0))
    n = nlev.value
    ab, as_ = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32)
    if n:
        ctx.call("fmk_aggregate_footprint", *args, ptr(ab), ptr(as_), c_i64(n))
    return np.arange(minl.value, minl.value + n, dtype=np.int32), ab, as_


def bucket_price_levels(all_price_levels: NDArray[np.int32], total_volumes: NDArray[np.float32],
                        n_bins: int) -> Tuple[NDArray[np.int32], NDArray[np.float32]]:
    """Reference volume.py:207-275 -> (binned_price_levels int32: bucket midpoints (+ the maximum for a leftover bucket),
    binned_volumes float32)."""
    pl = np.ascontiguousarray(all_price_levels, dtype=np.int32)
    v = np.ascontiguousarray(total_volumes, dtype=np.float32)
    if len(pl) != len(v):
        raise IndexError(f"index {min(len(pl), len(v))} is out of bounds for axis 0 with size {min(len(pl), len(v))}")
    ctx = _ffi.default_context()
    m = c_i64()
    ctx.call("fmk_bucket_price_levels", ptr(pl), ptr(v), c_i64(len(pl)), c_i64(int(n_bins)), None, None, c_i64(0), _byref(m))
    bl, bv = np.zeros(m.value, dtype=np.int32), np.zeros(m.value, dtype=np.float32)
    ctx.call("fmk_bucket_price_levels", ptr(pl), ptr(v), c_i64(len(pl)), c_i64(int(n_bins)), ptr(bl), ptr(bv), c_i64(m.value), _byref(m))
    return bl, bv


def comp_poc_hva_lva(price_levels: NDArray[np.int32], volumes: NDArray[np.float32], va_pct: float = 68.34) -> Tuple[int, int, int]:
    """Reference volume.py:278-365 -> (poc_price, hva_price, lva_price) in the units of `price_levels`.  Contract (include/fmk.h):
    the total is NumPy's pairwise float32 np.sum (the reference's pinned pure-Python mode), the walk's scalars are float64; the
    readings of the reference's scalars coincide unless the covered volume meets the threshold to the last float32 bit."""
    import ctypes as C
    pl = np.ascontiguousarray(price_levels, dtype=np.int32)
    v = np.ascontiguousarray(volumes, dtype=np.float32)
    if len(pl) != len(v):
        raise IndexError(f"index {min(len(pl), len(v))} is out of bounds for axis 0 with size {min(len(pl), len(v))}")
    poc, hva, lva = C.c_int32(), C.c_int32(), C.c_int32()
    _ffi.default_context().call("fmk_comp_poc_hva_lva", ptr(pl), ptr(v), c_i64(len(pl)), c_f64(va_pct), C.byref(poc),
                                C.byref(hva), C.byref(lva))
    return poc.value, hva.value, lva.value


def _byref(x):
    import ctypes as C
    return C.byref(x)


def calc_volume_percentage_above_poc(price_levels: NDArray[np.int32], volumes: NDArray[np.float32], poc_price: int) -> float:
    """Share (0..1) of the volume that sits on levels above `poc_price` (reference volume.py:367-391), on the device."""
    pl = np.ascontiguousarray(price_levels, dtype=np.int32)
    v = np.ascontiguousarray(volumes, dtype=np.float32)
    if len(pl) != len(v):
        raise IndexError(f"index {min(len(pl), len(v))} is out of bounds for axis 0 with size {min(len(pl), len(v))}")
    out = c_f64()
    import ctypes as C
    _ffi.default_context().call("fmk_calc_volume_percentage_above_poc", ptr(pl), ptr(v), c_i64(len(pl)),
                                C.c_int32(int(poc_price)), C.byref(out))
    return out.value


class VolumePro:
    """Rolling POC / value-area calculator (reference volume.py:12-131)."""

    def __init__(self, window_size: pd.Timedelta, n_bins: int = 27, va_pct: float = 68.34):
        self.window_size_sec = window_size.total_seconds()
        self.n_bins = n_bins
        self.va_pct = va_pct

    def reset_parameters(self, window_size_sec: int = None, n_bins: int = None, va_pct: float = None):
        self.window_size_sec = window_size_sec if window_size_sec is not None else self.window_size_sec
        self.n_bins = n_bins if n_bins is not None else self.n_bins
        self.va_pct = va_pct if va_pct is not None else self.va_pct

    def compute(self, bars: pd.DataFrame, fp_data) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """POC / HVA / LVA prices (NaN before the first full window) and the share of volume above the POC."""
        assert len(bars) == len(fp_data.bar_timestamps), "Bars and footprint data should have the same length."
        if getattr(fp_data, "level_offsets", None) is not None and getattr(fp_data, "flat", None) is not None:
            off = fp_data.level_offsets
            pl, bv, sv = (fp_data.flat[k] for k in ("price_levels", "buy_volumes", "sell_volumes"))
        else:
            off, pl, bv, sv = _to_csr(fp_data.price_levels, fp_data.buy_volumes, fp_data.sell_volumes)
        poc, hva, lva, pct = volume_profile_rolling_csr(fp_data.bar_timestamps, bars.high.values, bars.low.values, off,
                                                        pl, bv, sv, self.window_size_sec, self.n_bins,
                                                        fp_data.price_tick, self.va_pct)
        to_price = lambda a: np.where(a * fp_data.price_tick == 0, np.nan, a * fp_data.price_tick)   # volume.py:77-85
        return to_price(poc), to_price(hva), to_price(lva), pct

    def compute_range(self, bars: pd.DataFrame, fp_data, start: Union[str, int, pd.Timestamp],
                      end: Union[str, int, pd.Timestamp]):
        """`compute` on [start - window, end] (warm-up included), reference volume.py:89-131."""
        assert len(bars) == len(fp_data.bar_timestamps), "Bars and footprint data should have the same length."
        assert type(start) is type(end), "Start and end should be of the same type."
        if isinstance(start, int):
            end = pd.to_datetime(end)
        start = pd.to_datetime(start)
        adjusted_start = start - pd.Timedelta(seconds=self.window_size_sec)
        sub = fp_data[adjusted_start:end]
        bars_sub = bars.loc[pd.to_datetime(sub.bar_timestamps, unit="ns")]
        return (sub.bar_timestamps,) + self.compute(bars_sub, sub)
