"""Concrete bar kits: drop-in for finmlkit/bar/kit.py (same constructors, same `build_*` results).

Each kit only decides where bars close; the close indices are computed on the MI355X from the
builder's resident trade columns (BarBuilderBase._device) and handed to the shared reducers.
"""
from __future__ import annotations

import logging
from typing import Tuple

import numpy as np
import pandas as pd
from numpy.typing import NDArray

from .base import BarBuilderBase
from .data_model import TradesData
from .logic import _cusum_bar_indexer

logger = logging.getLogger(__name__)


class TimeBarKit(BarBuilderBase):
    """Fixed-interval time bars (reference kit.py:12-35)."""

    def __init__(self, trades: TradesData, period: pd.Timedelta):
        super().__init__(trades)
        self.interval = period.total_seconds()
        logger.info(f"Time bar builder initialized with interval: {self.interval} seconds.")

    def _comp_bar_close(self) -> Tuple[NDArray[np.int64], NDArray[np.int64]]:
        clock, idx = self._device().time_bar_index(self.interval)
        self._d_close_idx = idx
        return clock.to_host(), idx.to_host()

    def build_ohlcv(self) -> pd.DataFrame:
        """base.py:132-169 for time bars.  When the close indices have not been computed yet, the clock, the close indices and the
        OHLCV columns come from ONE library call (fmk_time_bars_ohlcv_dev: for 1-minute-sized bars one kernel launch); the values
        are those of _comp_bar_close() followed by the base class' build_ohlcv()."""
        if self._close_ts is None and self._close_indices is None:
            dev = self._device()
            logger.info("Calculating bar close tick indices and timestamps...")
            clock, idx, o = dev.time_bars_ohlcv(self.interval)
            if idx.n >= 2:
                self._d_close_idx = idx
                self._close_ts, self._close_indices = clock.to_host(), idx.to_host()
                return self._ohlcv_frame({k: v.to_host() for k, v in o.items()})
        return super().build_ohlcv()


class _ThresholdKit(BarBuilderBase):
    def _close_from(self, d_idx) -> Tuple[NDArray[np.int64], NDArray[np.int64]]:
        self._d_close_idx = d_idx
        close_ts = self._device().gather_ts(d_idx).to_host()      # timestamps[close_indices] (kit.py:66)
        return close_ts, d_idx.to_host()


class TickBarKit(_ThresholdKit):
    """Every `tick_count_thrs`-th trade closes a bar (reference kit.py:38-67)."""

    def __init__(self, trades: TradesData, tick_count_thrs: int):
        super().__init__(trades)
        self.tick_count_thrs = tick_count_thrs
        logger.info(f"Tick bar builder initialized with tick count: {tick_count_thrs}.")

    def _comp_bar_close(self):
        return self._close_from(self._device().tick_bar_index(self.tick_count_thrs))


class VolumeBarKit(_ThresholdKit):
    """Cumulative traded volume >= `volume_ths` closes a bar (reference kit.py:70-101)."""

    def __init__(self, trades: TradesData, volume_ths: float):
        super().__init__(trades)
        self.volume_ths = volume_ths
        logger.info(f"Volume bar builder initialized with volume: {volume_ths}.")

    def _comp_bar_close(self):
        return self._close_from(self._device().volume_bar_index(self.volume_ths))


class DollarBarKit(_ThresholdKit):
    """Cumulative price*volume >= `dollar_thrs` closes a bar, excess carried over (reference kit.py:104-137)."""

    def __init__(self, trades: TradesData, dollar_thrs: float):
        super().__init__(trades)
        self.dollar_thrs = dollar_thrs
        logger.info(f"Dollar bar builder initialized with dollar amount: {dollar_thrs}.")

    def _comp_bar_close(self):
        return self._close_from(self._device().dollar_bar_index(self.dollar_thrs))


class CUSUMBarKit(BarBuilderBase):
    """Symmetric CUSUM bars with an adaptive threshold sigma_mult * sigma (reference kit.py:140-181)."""

    def __init__(self, trades: TradesData, sigma, sigma_floor: float = 5e-4, sigma_mult: float = 2.):
        super().__init__(trades)
        self.lambda_mult, self._sigma, self.sigma_floor = sigma_mult, sigma, sigma_floor
        logger.info(f"CUSUM Bar builder initialized with: sigma multiplier={sigma_mult}.")

    def _comp_bar_close(self):
        """`_cusum_bar_indexer` (logic.py:152-221) on the builder's RESIDENT timestamp / price columns: only sigma is
        uploaded (and, forward-filled from its first valid entry, written back in place like the reference, :187-189)."""
        import ctypes as C
        from .._ffi import DeviceArray, c_f64, c_i64
        dev = self._device()
        n = dev.n
        sigma = self._sigma
        if not isinstance(sigma, np.ndarray) or len(sigma) != n or n == 0:
            # length quirks of the reference's chained comparison (logic.py:174-175): the NumPy-level function has them
            timestamps = self.trades_df["timestamp"].astype(np.int64).values
            close_indices = _cusum_bar_indexer(timestamps, self.trades_df["price"].values, sigma, self.sigma_floor,
                                               self.lambda_mult)
            return timestamps[close_indices], close_indices
        d_sigma = DeviceArray.from_host(dev.ctx, np.ascontiguousarray(sigma, dtype=np.float64))
        m = c_i64()
        args = (dev.ts.p, dev.price.p, d_sigma.p, c_i64(n), c_f64(self.sigma_floor), c_f64(self.lambda_mult))
        d_all = DeviceArray(dev.ctx, n, np.int64)                   # at most one close per tick: one pass, no count phase
        dev.ctx.call("fmk_cusum_bar_indexer_dev", *args, d_all.p, c_i64(n), C.byref(m), None)
        # the m closes move to a buffer of their own: a view would keep all n slots (8 B per tick) alive with the builder
        d_idx = DeviceArray(dev.ctx, m.value, np.int64)
        if m.value:
            dev.ctx.call("fmk_d2d", d_idx.p, d_all.p, C.c_size_t(m.value * 8))
        dev.ctx.sync()
        d_all.free()
        filled = d_sigma.to_host()
        if sigma.flags["WRITEABLE"]:
            sigma[...] = filled                                     # the in-place forward fill, visible to the caller
        else:
            self._sigma = filled                                    # read-only input: the builder keeps the filled copy (get_sigma)
        self._d_close_idx = d_idx
        return dev.gather_ts(d_idx).to_host(), d_idx.to_host()      # timestamps[close_indices] (kit.py:172-174)

    def get_sigma(self):
        """The (forward-filled) sigma at the close indices (reference kit.py:176-181)."""
        return self._sigma[self.bar_close_indices]
