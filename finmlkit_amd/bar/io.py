"""`TimeBarReader._resample` of finmlkit/bar/io.py (:890-950) on the MI355X: bars -> coarser bars.

The reference's HDF5 layer (`H5Inspector`, `AddTimeBarH5`, `TimeBarReader.read`: PyTables) is out of scope;
what is on the path either side of the tick->bar kernels is the re-aggregation of finished bars:
1-second bars -> `timeframe` bars with first / max / min / last, summed volume and trade count, volume-weighted VWAP and the
trades-weighted median of the per-bar medians.  The group keys are computed on the host with pandas' own `index.floor`
(so every timeframe string pandas accepts means the same thing); the aggregation runs on the device, one wave per group
(csrc/fmk_resample.hip), with pandas' arithmetic: Kahan-compensated sums in the column's dtype, NaN-skipping first / last /
max / min.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import pandas as pd

from .. import _ffi
from .._ffi import c_i64, ptr

_COLUMNS = ["open", "high", "low", "close", "volume", "trades", "vwap", "median_trade_size"]


def resample_bars(df: pd.DataFrame, timeframe: str) -> pd.DataFrame:
    """Reference: `TimeBarReader._resample(df, timeframe)` (io.py:890-950).  Same frame: columns open, high, low, close,
    volume (input dtype), trades, vwap (float32), median_trade_size (float32), indexed by the floored timestamps in order
    of first appearance, groups without any open dropped."""
    grouper = df.index.floor(timeframe)                                   # io.py:913
    codes, uniques = pd.factorize(grouper, sort=False)                    # groupby(..., sort=False): order of appearance
    n = len(df)
    cols = {c: df[c].values for c in _COLUMNS}
    if n and np.any(np.diff(codes) < 0):                                  # a key that re-appears later: make it contiguous
        order = np.argsort(codes, kind="stable")                          # rows keep their order inside a group
        codes = codes[order]
        cols = {c: v[order] for c, v in cols.items()}
    seg = np.concatenate([[0], np.flatnonzero(np.diff(codes)) + 1, [n]]).astype(np.int64) if n else np.zeros(1, np.int64)
    G = len(seg) - 1
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    vol = cols["volume"]
    vol = np.ascontiguousarray(vol, dtype=np.float32 if vol.dtype == np.float32 else np.float64)
    vw = cols["vwap"]
    vw = np.ascontiguousarray(vw, dtype=np.float32 if vw.dtype == np.float32 else np.float64)
    tr = np.ascontiguousarray(cols["trades"], dtype=np.int64)
    out = [np.empty(G, np.float64) for _ in range(4)] + [np.empty(G, vol.dtype), np.empty(G, np.int64),
                                                         np.empty(G, np.float32), np.empty(G, np.float32), np.empty(G, np.uint8)]
    if G:
        _ffi.default_context().call(
            "fmk_resample_bars", ptr(seg), c_i64(G), c_i64(n), ptr(f8(cols["open"])), ptr(f8(cols["high"])),
            ptr(f8(cols["low"])), ptr(f8(cols["close"])), ptr(vol), C.c_int(vol.dtype == np.float64), ptr(tr), ptr(vw),
            C.c_int(vw.dtype == np.float64), ptr(f8(cols["median_trade_size"])), *[ptr(a) for a in out])
    res = pd.DataFrame(dict(zip(_COLUMNS, out[:8])), index=uniques)
    res.index.name = df.index.name
    keep = out[8].astype(bool)
    return res if keep.all() else res[keep]                               # dropna(subset=["open"]) (io.py:948)


class TimeBarReader:
    """The resampling half of finmlkit.bar.io.TimeBarReader.  Reading bars from HDF5 (`read`, `list_keys`, ...) needs
    PyTables and is out of scope; frames from any source go through `_resample` exactly like the reference's."""

    def __init__(self, h5_path: str = None):
        self.h5_path = h5_path

    def _resample(self, df: pd.DataFrame, timeframe: str) -> pd.DataFrame:
        return resample_bars(df, timeframe)

    def read(self, *args, **kwargs):
        raise NotImplementedError("HDF5 bar storage (PyTables) is outside this build's scope; pass frames to _resample")
