"""Bar builder base class + per-bar reducers: drop-in for finmlkit/bar/base.py.

* `BarBuilderBase` (reference :24-300): same public surface (`build_ohlcv`,
  `build_directional_features`, `build_footprints`, `bar_close_indices`, `bar_close_timestamps`,
  abstract `_comp_bar_close`) and the same DataFrame / FootprintData schemas.  The trade columns are
  uploaded to HBM ONCE per builder and every `build_*` runs on the resident copy.
* module-level reducers (reference :306-850): same names, positional arguments, output tuples, dtypes
  and exceptions, computed by the HIP kernels of csrc/ through the host-pointer C ABI.
"""
from __future__ import annotations

import ctypes as C
import io
import logging
from abc import ABC, abstractmethod
from typing import Optional, Tuple

import numpy as np
import pandas as pd
from numpy.typing import NDArray

from .. import _ffi
from .._ffi import DeviceArray, c_f64, c_i64, ptr
from .data_model import FootprintData, TradesData
from .utils import comp_price_tick_size

logger = logging.getLogger(__name__)


class BarBuilderBase(ABC):
    """Template for bar samplers: subclasses decide where bars close (`_comp_bar_close`), the base class
    aggregates ticks between closes.  Mirrors finmlkit.bar.base.BarBuilderBase."""

    def __init__(self, trades: TradesData):
        self.trades_df = trades.data
        self._close_ts: Optional[NDArray[np.int64]] = None
        self._close_indices: Optional[NDArray[np.int64]] = None
        self._highs: Optional[NDArray[np.float64]] = None
        self._lows: Optional[NDArray[np.float64]] = None
        self._dev = None           # engine.DeviceTrades (lazy)
        self._d_close_idx = None   # DeviceArray mirror of _close_indices

    def __str__(self) -> str:
        members = "\n".join(f"{k}: {v}" for k, v in self.__dict__.items() if not k.startswith("_d"))
        buf = io.StringIO()
        try:
            self.trades_df.info(buf=buf)
            info = buf.getvalue()
        except Exception:
            info = "<unavailable>"
        return f"Class: {self.__class__.__name__} with members:\n{members}\nRaw trades data:\n{info}"

    # ------------------------------------------------------------------ device residency
    def _device(self):
        """Upload timestamp / price / amount (/ side) once; later builds reuse the HBM copy."""
        if self._dev is None:
            from ..engine import DeviceTrades
            df = self.trades_df
            # .values of a frame column is a view; from_numpy converts only what is not already int64 / float64 / float32 / int8
            side = df["side"].values if "side" in df.columns else None
            # the side column travels when a builder first needs it: build_ohlcv (the reference's published benchmark) does not
            self._dev = DeviceTrades.from_numpy(df["timestamp"].values, df["price"].values, df["amount"].values, side,
                                                lazy_side=True)
        return self._dev

    @abstractmethod
    def _comp_bar_close(self) -> Tuple[NDArray[np.int64], NDArray[np.int64]]:
        """Return (close timestamps, close indices), first entry = open edge (reference base.py:92-99)."""

    def _set_bar_close(self):
        if self._close_ts is None and self._close_indices is None:
            logger.info("Calculating bar close tick indices and timestamps...")
            self._close_ts, self._close_indices = self._comp_bar_close()
        if self._d_close_idx is None:
            self._d_close_idx = DeviceArray.from_host(self._device().ctx,
                                                      np.ascontiguousarray(self._close_indices, dtype=np.int64))

    @property
    def bar_close_indices(self) -> Optional[NDArray[np.int64]]:
        if self._close_indices is None:
            self._set_bar_close()
        return self._close_indices[1:]

    @property
    def bar_close_timestamps(self) -> Optional[NDArray[np.int64]]:
        if self._close_ts is None:
            self._set_bar_close()
        return self._close_ts[1:]

    def _check_indices(self):
        if len(self._close_indices) < 2:
            raise ValueError("Bar close indices must contain at least two elements.")

    # ------------------------------------------------------------------ builders
    def build_ohlcv(self) -> pd.DataFrame:
        """OHLCV + VWAP + trade count + median trade size per bar (reference base.py:132-169)."""
        from ..engine import to_host
        self._set_bar_close()
        self._check_indices()
        return self._ohlcv_frame(to_host(self._device().bar_ohlcv(self._d_close_idx)))

    def _ohlcv_frame(self, o) -> pd.DataFrame:
        """The frame of base.py:148-169 from the host copies of the eight OHLCV columns."""
        # the frame shares memory with these two (copy=False below; pandas 2 without copy-on-write), and build_footprints / the volume
        # profile read them later: the kit keeps copies of its own, so an in-place edit of the returned frame changes nothing here --
        # like the reference's frame, which is an independent copy (base.py:148-169)
        self._highs, self._lows = o["high"].copy(), o["low"].copy()
        # (the same frame as base.py:148-169 -- columns, dtypes, a DatetimeIndex named "timestamp", its freq for time bars -- built
        #  index first and without the detour through an int64 column + to_datetime + set_index: 0.5 instead of 3 ms for 44 640 bars,
        #  a fifth of what TimeBarKit.build_ohlcv() costs on 39 M host-resident trades beyond the upload itself)
        ts = np.ascontiguousarray(self.bar_close_timestamps, dtype=np.int64)
        try:
            idx = pd.DatetimeIndex(ts.view("M8[ns]"), name="timestamp",
                                   freq=pd.Timedelta(seconds=self.interval) if hasattr(self, "interval") else None)
        except ValueError:                                               # (a clock the freq does not fit: the reference's own assignment decides)
            idx = pd.DatetimeIndex(ts.view("M8[ns]"), name="timestamp")
            if hasattr(self, "interval"):
                idx.freq = pd.Timedelta(seconds=self.interval)
        return pd.DataFrame({"open": o["open"], "high": o["high"], "low": o["low"], "close": o["close"], "volume": o["volume"],
                             "trades": o["trades"], "median_trade_size": o["median_trade_size"], "vwap": o["vwap"]}, index=idx, copy=False)

    def build_directional_features(self) -> pd.DataFrame:
        """Order-flow features per bar (reference base.py:171-212)."""
        from ..engine import to_host
        self._set_bar_close()             # fewer than two close indices = no bars: empty frame (only comp_bar_ohlcv
        if "side" not in self.trades_df.columns:          # checks for that in the reference, base.py:334-335)
            raise KeyError("side")
        dev = self._device()
        out, nz = dev.bar_directional(self._d_close_idx)
        d = to_host(out)
        if int(nz.to_host()[0]) > 0:     # reference: cum_spread / (buy + sell) with no signed tick (base.py:536)
            raise ZeroDivisionError("division by zero")
        df = pd.DataFrame({"timestamp": self.bar_close_timestamps})
        for k in ("ticks_buy", "ticks_sell", "volume_buy", "volume_sell", "dollars_buy", "dollars_sell",
                  "mean_spread", "max_spread", "cum_ticks_min", "cum_ticks_max"):
            df[k] = d[k]
        df["cum_volume_min"], df["cum_volume_max"] = d["cum_volumes_min"], d["cum_volumes_max"]
        df["cum_dollars_min"], df["cum_dollars_max"] = d["cum_dollars_min"], d["cum_dollars_max"]
        df["timestamp"] = pd.to_datetime(df["timestamp"], unit="ns")
        df.set_index("timestamp", inplace=True)
        return df

    def build_trade_size_features(self, theta, theta_mult: float = 5.0) -> pd.DataFrame:
        """Relative mean / 95th-percentile trade size, block share and size Gini per bar
        (reference base.py:214-245)."""
        self._set_bar_close()
        nb = max(len(self._close_indices) - 1, 0)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if len(theta) != nb:
            raise ValueError("Theta should match the the number of bars (len(bar_close_indices) - 1).")
        out = self._device().bar_trade_size(self._d_close_idx, theta, theta_mult)
        df = pd.DataFrame({"timestamp": self.bar_close_timestamps, **out})
        df["timestamp"] = pd.to_datetime(df["timestamp"], unit="ns")
        df.set_index("timestamp", inplace=True)
        return df

    def build_footprints(self, price_tick_size=None, imbalance_factor=3.0) -> FootprintData:
        """Per-bar price-level footprints + imbalance statistics (reference base.py:247-300)."""
        from ..engine import to_host
        self._set_bar_close()
        if self._highs is None or self._lows is None:
            self.build_ohlcv()            # raises for fewer than two close indices, as the reference does here
        if price_tick_size is None:
            price_tick_size = comp_price_tick_size(self.trades_df["price"].values)
        logger.info(f"Price tick size is set to: {price_tick_size}")
        if "side" not in self.trades_df.columns:
            raise KeyError("side")
        dev = self._device()
        lows = DeviceArray.from_host(dev.ctx, self._lows)
        highs = DeviceArray.from_host(dev.ctx, self._highs)
        off, flat, bar, bad = dev.bar_footprints(self._d_close_idx, lows, highs, price_tick_size, imbalance_factor)
        if int(bad.to_host()[0]) > 0:
            raise ValueError("Something went wrong! Invalid price level index!")
        fp = FootprintData.from_csr(self.bar_close_timestamps, price_tick_size, off.to_host(), to_host(flat),
                                    to_host(bar))
        fp.cast_to_numba_list()
        return fp


# --------------------------------------------------------------------------------------------
# CORE FUNCTIONS (NumPy in / NumPy out)
# --------------------------------------------------------------------------------------------
def _check_close_indices(ci: np.ndarray, n: int):
    """The reducers index ticks ci[i] + 1 .. ci[i + 1] directly (base.py:349-391 and siblings).  An index at or past the end
    of the arrays is an IndexError in the reference's Python mode and an out-of-bounds read under Numba; here it would be an
    out-of-bounds DEVICE read, so the NumPy-facing functions refuse it.  -1 (first bar opens at tick 0) is the smallest
    meaningful entry.  comp_bar_trade_size_features is exempt: it slices, and a slice clamps (base.py:590)."""
    if len(ci) < 2:
        return                                   # no bars, nothing is indexed (one element: the reference returns empties)
    if int(ci[1:].max()) >= n or int(ci.min()) < -1:
        bad = int(ci[1:].max()) if int(ci[1:].max()) >= n else int(ci.min())
        raise IndexError(f"index {bad} is out of bounds for axis 0 with size {n}")


def comp_bar_ohlcv(prices: NDArray[np.float64], volumes: NDArray, bar_close_indices: NDArray[np.int64]):
    """Reference: finmlkit/bar/base.py:306-407.

    Returns (open, high, low, close, volume[f32], vwap, trades[i64], median_trade_size)."""
    if len(prices) != len(volumes):
        raise ValueError("Prices and volumes arrays must have the same length.")
    if len(bar_close_indices) < 2:
        raise ValueError("Bar close indices must contain at least two elements.")
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f64 = _ffi.amount_array(volumes)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    _check_close_indices(ci, len(p))
    ctx = _ffi.default_context()
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
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f64 = _ffi.amount_array(volumes)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    sd = np.ascontiguousarray(trade_sides, dtype=np.int8)
    _check_close_indices(ci, len(p))
    ctx = _ffi.default_context()
    # no length check on bar_close_indices in the reference (only comp_bar_ohlcv has one, base.py:334-335): one element
    # -> zero bars, empty outputs; none -> NumPy's "negative dimensions are not allowed" from the allocation below
    nb = len(ci) - 1
    outs = {k: np.empty(nb, dt) for k, dt in _ffi.DIRECTIONAL_FIELDS}
    st = _ffi.DirectionalOut(**{k: a.ctypes.data for k, a in outs.items()})
    ctx.call("fmk_comp_bar_directional", ptr(p), ptr(v), C.c_int(f64), c_i64(len(p)), ptr(ci), c_i64(len(ci)),
             ptr(sd), C.byref(st))
    return tuple(outs[k] for k, _ in _ffi.DIRECTIONAL_FIELDS)


def comp_bar_trade_size_features(amounts: NDArray, theta: NDArray[np.float64], bar_close_indices: NDArray[np.int64],
                                 theta_mult: float):
    """Reference: finmlkit/bar/base.py:549-612.

    Returns float32 arrays (mean_size_rel, size_95_rel, pct_block, size_gini); NaN where the reference
    leaves NaN (empty bar, theta == 0, zero total volume)."""
    ctx = _ffi.default_context()
    v, f64 = _ffi.amount_array(amounts)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    if len(th) != len(ci) - 1:
        raise ValueError("Theta should match the the number of bars (len(bar_close_indices) - 1).")
    nb = len(ci) - 1
    outs = tuple(np.empty(nb, np.float32) for _ in range(4))
    ctx.call("fmk_comp_bar_trade_size", ptr(v), C.c_int(f64), c_i64(len(v)), ptr(th), ptr(ci), c_i64(len(ci)),
             c_f64(theta_mult), *[ptr(o) for o in outs])
    return outs


def comp_bar_footprints_csr(prices, amounts, bar_close_indices, trade_sides, price_tick_size, bar_lows,
                            bar_highs, imbalance_factor):
    """CSR form of comp_bar_footprints: (level_offsets[B+1], flat per-level dict, per-bar dict)."""
    p = np.ascontiguousarray(prices, dtype=np.float64)
    v, f64 = _ffi.amount_array(amounts)
    ci = np.ascontiguousarray(bar_close_indices, dtype=np.int64)
    sd = np.ascontiguousarray(trade_sides, dtype=np.int8)
    lo = np.ascontiguousarray(bar_lows, dtype=np.float64)
    hi = np.ascontiguousarray(bar_highs, dtype=np.float64)
    _check_close_indices(ci, len(p))
    ctx = _ffi.default_context()
    # like the reference (base.py:615-752): no length check; one element -> zero bars -> empty lists / arrays
    # (tests/bars/test_comp_bar_footprints.py::test_comp_bar_footprints_empty_bar of the reference)
    nb = len(ci) - 1
    bar = {k: np.empty(nb, dt) for k, dt in _ffi.FOOTPRINT_BAR_FIELDS}     # nb == -1: NumPy's ValueError
    off = np.empty(nb + 1, np.int64)
    args = (ptr(p), ptr(v), C.c_int(f64), c_i64(len(p)), ptr(ci), c_i64(len(ci)), ptr(sd),
            c_f64(price_tick_size), ptr(lo), ptr(hi), c_f64(imbalance_factor), ptr(off))
    ctx.call("fmk_comp_bar_footprints", *args, None)
    tot = int(off[-1])
    flat = {k: np.empty(tot, dt) for k, dt in _ffi.FOOTPRINT_FLAT_FIELDS}
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


def comp_footprint_features(price_levels, buy_volumes, sell_volumes, imbalance_multiplier):
    """Reference: finmlkit/bar/base.py:755-850, for ONE bar's level arrays.

    Evaluated by the same device code as comp_bar_footprints: the level profile is replayed as a
    synthetic one-bar tick stream (one buy and one sell tick per level carrying that level's volume)."""
    lv = np.ascontiguousarray(price_levels, dtype=np.int32)
    b = np.ascontiguousarray(buy_volumes, dtype=np.float32)
    s = np.ascontiguousarray(sell_volumes, dtype=np.float32)
    L = len(lv)
    if L == 0:
        raise ValueError("attempt to get argmax of an empty sequence")
    if np.any(np.diff(lv) != 1):
        raise ValueError("comp_footprint_features: price_levels must be consecutive integers")
    px = np.repeat(lv.astype(np.float64), 2)                 # tick size 1.0: level == price
    am = np.stack([b, s], axis=1).reshape(-1)
    sd = np.tile(np.array([1, -1], dtype=np.int8), L)
    ci = np.array([-1, 2 * L - 1], dtype=np.int64)
    off, flat, bar = comp_bar_footprints_csr(px, am, ci, sd, 1.0, np.array([float(lv[0])]),
                                             np.array([float(lv[-1])]), imbalance_multiplier)
    return (flat["buy_imbalances"].view(np.bool_), flat["sell_imbalances"].view(np.bool_),
            int(bar["imb_max_run_signed"][0]), np.int32(bar["cot_price_levels"][0]), float(bar["vp_skew"][0]),
            float(bar["vp_gini"][0]))
