"""Host-side containers of the bar path: `TradesData` and `FootprintData`.

Same constructor arguments, attributes and column schema as finmlkit/bar/data_model.py
(TradesData :121-252, FootprintData :775-1058) for everything the bar builders touch, including the
`preprocess=True` pipeline (:236-246, 296-400): unit conversion, id sort + integrity report and the
re-sort stay pandas (host), the two sequential loops -- split-trade merge and tick-rule side
inference -- run on the MI355X (finmlkit_amd.bar.utils).  Not here (out of scope):
the HDF5 store.
"""
from __future__ import annotations

import datetime as dt
from dataclasses import dataclass
from typing import List, Optional, Union

import numpy as np
import pandas as pd
from numpy.typing import NDArray

from .utils import comp_trade_side_vector, footprint_to_dataframe, merge_split_trades

_NS_PER_UNIT = {"s": 1_000_000_000, "ms": 1_000_000, "us": 1_000, "ns": 1}


class TradesData:
    """Raw trades as a DataFrame (`timestamp` int64 ns, `price` f64, `amount`, `id`, optional `side`)
    indexed by datetime -- the object every bar kit is constructed from."""

    def __init__(self, ts: NDArray, px: NDArray, qty: NDArray, id: NDArray = None, *,
                 is_buyer_maker: NDArray = None, side=None, dt_index: Optional[pd.DatetimeIndex] = None,
                 timestamp_unit: Optional[str] = None, preprocess: bool = False, proc_res: Optional[str] = None,
                 name=None):
        for label, arr, optional in (("ts", ts, False), ("px", px, False), ("qty", qty, False), ("id", id, True)):
            if not (optional and arr is None) and not isinstance(arr, np.ndarray):
                raise TypeError(f"{label} must be a np.ndarray")
        if is_buyer_maker is not None and not isinstance(is_buyer_maker, np.ndarray):
            raise TypeError("is_buyer_maker must be None or np.ndarray")
        if side is not None and not isinstance(side, np.ndarray):
            raise TypeError("side must be None or np.ndarray")
        self._start_date = self._end_date = None
        self._data = pd.DataFrame({"timestamp": ts, "price": px, "amount": qty, "id": id})
        self.is_buyer_maker = is_buyer_maker
        if side is not None:
            self._data["side"] = side
        self._orig_timestamp_unit = timestamp_unit if timestamp_unit else self._infer_timestamp_unit()
        self.name = name
        self.missing_pct = 0
        self.data_ok = None
        self.discontinuities = []
        if preprocess:                               # data_model.py:236-246 of the reference
            if id is None:
                raise ValueError("id is required if preprocess is True")
            self._convert_timestamps_to_ns()
            self._sort_trades()
            self._merge_trades()
            self._apply_timestamp_resolution(proc_res)
            if "side" not in self._data.columns:
                self._data["side"] = comp_trade_side_vector(self._data["price"].values)
        if dt_index is not None:
            self._data.set_index(dt_index, inplace=True)
        else:
            self._data.set_index(pd.to_datetime(self._data["timestamp"], unit="ns"), inplace=True)
            self._data.index.name = "datetime"

    @property
    def start_date(self):
        return self._start_date

    @property
    def end_date(self):
        return self._end_date

    def set_view_range(self, start, end):
        """Restrict `.data` to [start, end] (datetime-index slice), like the reference."""
        start = pd.Timestamp(start) if isinstance(start, str) else start
        end = pd.Timestamp(end) if isinstance(end, str) else end
        if start >= end:
            raise ValueError("Start timestamp must be before end timestamp.")
        self._start_date, self._end_date = start, end

    @property
    def data(self) -> pd.DataFrame:
        if self._start_date is None and self._end_date is None:
            return self._data
        return self._data.loc[self._start_date: self._end_date]

    @property
    def orig_timestamp_unit(self) -> str:
        return self._orig_timestamp_unit

    # ------------------------------------------------------------------ preprocess=True pipeline
    def _convert_timestamps_to_ns(self) -> None:
        unit = self._orig_timestamp_unit
        if unit not in _NS_PER_UNIT:
            raise ValueError(f"Invalid timestamp format! Must be one of: {', '.join(_NS_PER_UNIT)}")
        self._data["timestamp"] = np.multiply(self._data["timestamp"].values, _NS_PER_UNIT[unit], dtype=np.int64)

    def _validate_data(self) -> None:
        """Report gaps in the (sorted, unique) trade ids; gaps that also span more than a minute are recorded in
        `discontinuities` and clear `data_ok` (reference data_model.py:253-293)."""
        ids = self._data["id"].values
        gaps = np.flatnonzero(np.diff(ids) > 1)
        if len(gaps) == 0:
            return
        ts = self._data["timestamp"].values
        missing = (ids[gaps + 1] - ids[gaps] - 1).astype(np.int64)
        for g, miss in zip(gaps, missing):
            before, after = pd.to_datetime(int(ts[g]), unit="ns"), pd.to_datetime(int(ts[g + 1]), unit="ns")
            if after - before > pd.Timedelta(minutes=1):
                self.data_ok = False
                self.discontinuities.append({"start_id": int(ids[g]), "end_id": int(ids[g + 1]),
                                             "missing_ids": int(miss), "pre_gap_time": before,
                                             "post_gap_time": after, "time_interval": after - before})
        self.missing_pct = int(missing.sum()) / len(self._data) * 100

    def _sort_trades(self) -> None:
        self.data_ok = True
        self.discontinuities = []
        self._data.sort_values(by=["id"], inplace=True)
        self._data.reset_index(drop=True, inplace=True)
        if self._data["id"].duplicated().any():
            self._data.drop_duplicates(subset="id", keep="first", inplace=True)
            self.data_ok = False
        self._validate_data()
        if not self._data["timestamp"].is_monotonic_increasing:
            self._data.sort_values(by=["timestamp", "id"], inplace=True)
        self._data.reset_index(drop=True, inplace=True)

    def _merge_trades(self) -> None:
        """Split-trade merge on the device; like the reference the merged frame has no `id` column and
        `is_buyer_maker` is used as given (data_model.py:324-342)."""
        ts, px, am, side = merge_split_trades(self._data["timestamp"].values.astype(np.int64),
                                              self._data["price"].values.astype(np.float64),
                                              self._data["amount"].values.astype(np.float32), self.is_buyer_maker)
        self._data = pd.DataFrame({"timestamp": ts, "price": px, "amount": am})
        if self.is_buyer_maker is not None:
            self._data["side"] = side

    def _apply_timestamp_resolution(self, proc_res: Optional[str]) -> None:
        if proc_res and proc_res != self._orig_timestamp_unit:
            if proc_res not in _NS_PER_UNIT:
                raise ValueError(f"Invalid processing resolution: {proc_res}. Must be one of: "
                                 f"{', '.join(_NS_PER_UNIT)}")
            res = _NS_PER_UNIT[proc_res]
            self._data["timestamp"] = (self._data["timestamp"] // res) * res

    def _infer_timestamp_unit(self) -> str:
        max_ts = self._data["timestamp"].max()
        if max_ts > 1e18:
            return "ns"
        if max_ts > 1e15:
            return "us"
        if max_ts > 1e12:
            return "ms"
        return "s"


_LISTS = ("price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks", "buy_imbalances",
          "sell_imbalances")
_PER_BAR = ("cot_price_levels", "sell_imbalances_sum", "buy_imbalances_sum", "imb_max_run_signed", "vp_skew",
            "vp_gini")


@dataclass
class FootprintData:
    """Per-bar price-level footprints: seven ragged columns (one array per bar) + per-bar statistics.

    Field names / dtypes follow finmlkit/bar/data_model.py:797-813.  When produced by the HIP path the
    per-bar arrays are zero-copy views into CSR buffers; `level_offsets` (int64[B+1]) and `flat`
    (dict of the contiguous per-level arrays) expose that layout for consumers that want it."""
    bar_timestamps: NDArray[np.int64]
    price_tick: float
    price_levels: Union[list, NDArray]
    buy_volumes: Union[list, NDArray]
    sell_volumes: Union[list, NDArray]
    buy_ticks: Union[list, NDArray]
    sell_ticks: Union[list, NDArray]
    buy_imbalances: Union[list, NDArray]
    sell_imbalances: Union[list, NDArray]
    cot_price_levels: Optional[NDArray[np.int32]] = None
    sell_imbalances_sum: Optional[NDArray[np.uint16]] = None
    buy_imbalances_sum: Optional[NDArray[np.uint16]] = None
    imb_max_run_signed: Optional[NDArray[np.int16]] = None
    vp_skew: Optional[NDArray[np.float64]] = None
    vp_gini: Optional[NDArray[np.float64]] = None
    level_offsets: Optional[NDArray[np.int64]] = None
    flat: Optional[dict] = None
    _datetime_index: pd.Series = None

    def __post_init__(self):
        self._datetime_index = pd.to_datetime(self.bar_timestamps, unit="ns")

    def __len__(self) -> int:
        return len(self.bar_timestamps)

    def __repr__(self) -> str:
        rng = f"{self._datetime_index[0]} to {self._datetime_index[-1]}" if len(self) else "empty"
        present = lambda a: "present" if a is not None else "missing"
        return (f"FootprintData:\n  Number of Bars: {len(self)}\n  Price Tick: {self.price_tick}\n"
                f"  Date Range: {rng}\n  Array Types: {type(self.price_levels).__name__}\n"
                f"  COT Price Levels: {present(self.cot_price_levels)}\n"
                f"  Imbalance sums: {present(self.buy_imbalances_sum)}\n"
                f"  VP Skew / Gini: {present(self.vp_skew)} / {present(self.vp_gini)}\n"
                f"  Total Memory Usage: {self.memory_usage():.3f} MB\n")

    def __getitem__(self, key) -> "FootprintData":
        if isinstance(key, slice) and isinstance(key.start, (str, dt.datetime)) and \
                isinstance(key.stop, (str, dt.datetime)):
            a, b = self._datetime_index.slice_locs(start=key.start, end=key.stop)
            return self[a:b]
        if not isinstance(key, (slice, int)):
            raise TypeError("Invalid argument type. Expected a slice or integer index.")
        opt = lambda a: a[key] if a is not None else None
        return FootprintData(
            bar_timestamps=self.bar_timestamps[key], price_tick=self.price_tick,
            **{k: getattr(self, k)[key] for k in _LISTS}, **{k: opt(getattr(self, k)) for k in _PER_BAR})

    @classmethod
    def from_csr(cls, bar_timestamps, price_tick, level_offsets, flat: dict, per_bar: dict) -> "FootprintData":
        """Wrap the CSR output of comp_bar_footprints_csr without copying the level data."""
        nb = len(level_offsets) - 1
        off = level_offsets

        def split(a):
            return [a[off[i]:off[i + 1]] for i in range(nb)]
        lists = {k: split(flat[k].view(np.bool_) if flat[k].dtype == np.uint8 else flat[k]) for k in _LISTS}
        return cls(bar_timestamps=np.asarray(bar_timestamps, dtype=np.int64), price_tick=price_tick, **lists,
                   **{k: per_bar[k] for k in _PER_BAR}, level_offsets=off, flat=flat)

    @classmethod
    def from_numba(cls, data, price_tick: float) -> "FootprintData":
        """Same positional layout as the reference's `from_numba` (data_model.py:896-927)."""
        inst = cls(bar_timestamps=np.array(data[0], dtype=np.int64), price_tick=price_tick,
                   price_levels=np.array(data[1], dtype=object), buy_volumes=np.array(data[2], dtype=object),
                   sell_volumes=np.array(data[3], dtype=object), buy_ticks=np.array(data[4], dtype=object),
                   sell_ticks=np.array(data[5], dtype=object), buy_imbalances=np.array(data[6], dtype=object),
                   sell_imbalances=np.array(data[7], dtype=object),
                   buy_imbalances_sum=np.array(data[8], dtype=np.uint16),
                   sell_imbalances_sum=np.array(data[9], dtype=np.uint16),
                   cot_price_levels=np.array(data[10], dtype=np.int32),
                   imb_max_run_signed=np.array(data[11], dtype=np.int16),
                   vp_skew=np.array(data[12], dtype=np.float64), vp_gini=np.array(data[13], dtype=np.float64))
        if not inst.is_valid():
            raise ValueError("Inconsistent data length in the FootprintData container!")
        return inst

    def get_df(self) -> pd.DataFrame:
        return footprint_to_dataframe(self.bar_timestamps, self.price_levels, self.buy_volumes, self.sell_volumes,
                                      self.buy_ticks, self.sell_ticks, self.buy_imbalances, self.sell_imbalances,
                                      self.price_tick)

    def cast_to_numba_list(self):
        """The reference converts to numba.typed.List here; without Numba a plain list is the equivalent."""
        for k in _LISTS:
            setattr(self, k, list(getattr(self, k)))

    def cast_to_numpy(self):
        for k in _LISTS:
            src = getattr(self, k)
            arr = np.empty(len(src), dtype=object)
            for i, a in enumerate(src):
                arr[i] = a
            setattr(self, k, arr)

    def memory_usage(self) -> float:
        """Approximate footprint in MB (array payloads only)."""
        total = self.bar_timestamps.nbytes
        for k in _LISTS:
            total += sum(np.asarray(a).nbytes for a in getattr(self, k))
        for k in _PER_BAR:
            a = getattr(self, k)
            total += a.nbytes if a is not None else 0
        return total / (1024 ** 2)

    def is_valid(self) -> bool:
        n = len(self.bar_timestamps)
        return all(len(getattr(self, k)) == n for k in _LISTS)
