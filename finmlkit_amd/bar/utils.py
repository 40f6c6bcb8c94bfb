"""Helpers of the bar builders (the counterparts of finmlkit/bar/utils.py that the path calls): the
tick-size estimate (pure NumPy on at most 10k samples), the footprint DataFrame view, and the two loops
of `TradesData(preprocess=True)` -- `merge_split_trades`, `comp_trade_side_vector` -- on the MI355X."""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import pandas as pd
from numpy.typing import NDArray


def merge_split_trades(timestamps: NDArray[np.int64], prices: NDArray[np.float64], amounts: NDArray[np.float32],
                       is_buyer_maker: Optional[NDArray[np.bool_]]):
    """Merge split trades: same timestamp, maker flag and price (|dp| < 1e-8 against the merged trade's
    first price); amounts summed in float32 in trade order.  Reference: finmlkit/bar/utils.py:263-329.

    -> (timestamps, prices, amounts float32, side int8) -- side is an empty array without `is_buyer_maker`."""
    import ctypes as C

    from .. import _ffi
    from .._ffi import c_i64, ptr
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    px = np.ascontiguousarray(prices, dtype=np.float64)
    am = np.ascontiguousarray(amounts, dtype=np.float32)
    n = len(ts)
    if not (len(px) == len(am) == n) or n == 0:
        raise ValueError("timestamps, prices and amounts must be non-empty arrays of one length")
    ibm = None if is_buyer_maker is None else np.ascontiguousarray(is_buyer_maker, dtype=np.uint8)
    o_ts, o_px, o_am = np.empty(n, np.int64), np.empty(n, np.float64), np.empty(n, np.float32)
    o_sd = np.empty(n if ibm is not None else 0, np.int8)
    m = c_i64()
    _ffi.default_context().call("fmk_merge_split_trades", ptr(ts), ptr(px), ptr(am), None if ibm is None else ptr(ibm),
                                c_i64(n), ptr(o_ts), ptr(o_px), ptr(o_am), ptr(o_sd) if ibm is not None else None,
                                c_i64(n), C.byref(m))
    k = m.value
    return o_ts[:k], o_px[:k], o_am[:k], (o_sd[:k] if ibm is not None else np.empty(0, dtype=np.int8))


def comp_trade_side_vector(prices: NDArray[np.float64]) -> NDArray[np.int8]:
    """Tick rule: +1 / -1 after an up / down move larger than 1e-12, else the previous side; side[0] = 0.
    Reference: finmlkit/bar/utils.py:26-46."""
    from .. import _ffi
    from .._ffi import c_i64, ptr
    px = np.ascontiguousarray(prices, dtype=np.float64)
    out = np.empty(len(px), np.int8)
    _ffi.default_context().call("fmk_comp_trade_side_vector", ptr(px), c_i64(len(px)), ptr(out))
    return out


def comp_price_tick_size(prices: NDArray[np.float64]) -> float:
    """Smallest price increment of a price sample (reference: finmlkit/bar/utils.py:49-81).

    Called by `build_footprints(price_tick_size=None)` on the first 10,000 prices: unique values
    (rounded to 12 decimals) are scaled to integers by the decade of the smallest gap and the
    tick is the gcd of the integer gaps.  Returns 0.0 if the sample holds a single price."""
    if len(prices) == 0:
        raise ValueError("Empty prices array")
    sample = np.round(np.asarray(prices[:min(10000, len(prices))], dtype=np.float64), decimals=12)
    uniq = np.unique(sample)
    if len(uniq) <= 1:
        return 0.0
    gaps = np.diff(uniq)
    scale = 10.0 ** (-np.floor(np.log10(np.min(gaps[gaps > 0]))))
    ints = np.round(uniq * scale).astype(np.int64)
    tick = 0
    for g in np.diff(ints):
        g = int(g)
        if g > 0:
            tick = g if tick == 0 else math.gcd(tick, g)
            if tick == 1:
                break
    return tick / scale


def footprint_to_dataframe(bar_timestamps, price_levels, buy_volumes, sell_volumes, buy_ticks, sell_ticks,
                           buy_imbalance, sell_imbalance, price_tick) -> pd.DataFrame:
    """Long-format footprint table (schema of finmlkit/bar/utils.py:129-209): MultiIndex
    (bar_idx, bar_datetime_idx), one row per (bar, price level), price descending inside a bar."""
    n_bars = len(price_levels)
    lens = np.fromiter((len(x) for x in price_levels), dtype=np.int64, count=n_bars)
    bar_ids = np.repeat(np.arange(n_bars), lens)
    bar_dt = pd.to_datetime(np.asarray(bar_timestamps))

    def cat(parts, dtype=None):
        return np.concatenate([np.asarray(p) for p in parts]) if n_bars else np.zeros(0, dtype=dtype)

    df = pd.DataFrame({
        "price_level": cat(price_levels, np.int32),
        "sell_ticks": cat(sell_ticks, np.int32),
        "buy_ticks": cat(buy_ticks, np.int32),
        "sell_volume": cat(sell_volumes, np.float32),
        "buy_volume": cat(buy_volumes, np.float32),
        "sell_imbalance": cat(sell_imbalance, bool),
        "buy_imbalance": cat(buy_imbalance, bool),
    }, index=pd.MultiIndex.from_arrays([bar_ids, bar_dt[bar_ids]], names=["bar_idx", "bar_datetime_idx"]))
    df["price_level"] = df["price_level"] * price_tick
    return df.sort_values(by=["bar_datetime_idx", "price_level"], ascending=[True, False])
