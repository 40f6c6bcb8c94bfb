"""Host-side helpers of the bar builders (the counterparts of finmlkit/bar/utils.py that the hot
path actually calls).  Pure NumPy on at most 10k samples / on already-reduced bar data."""
from __future__ import annotations

import math

import numpy as np
import pandas as pd
from numpy.typing import NDArray


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
