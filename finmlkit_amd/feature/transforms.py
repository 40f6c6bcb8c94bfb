"""The tick-level transforms of the hot path: `ReturnT`, `EWMST`, `RealizedVolatility` (+ `Compose`).

Counterparts of finmlkit/feature/transforms.py:89-117 (ReturnT), :308-332 (EWMST) and the
pipeline part of finmlkit/feature/kit.py:Compose (:630-720), enough to run the QuickStart flow
    Compose(ReturnT(window, input_col="price"), EWMST(half_life))(trades.data)
on the MI355X.  The reference's `backend="nb"` (Numba) and `"pd"` both map to the HIP path here
(the reference's own `_pd` of these two transforms already delegates to the Numba kernel).
The other ~37 bar-level transforms of the reference are out of scope.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import pandas as pd

from .. import _ffi
from .._ffi import DeviceArray, c_f64, c_i64
from .core.utils import comp_lagged_returns
from .core.volatility import ewmst, realized_vol


class SISOTransform:
    """Single-input single-output transform: output column = f"{input_col}_{output_col}"."""

    def __init__(self, input_col: str, output_col: str):
        self.requires = [input_col]
        self.produces = [output_col]

    @property
    def output_name(self) -> str:
        return f"{self.requires[0]}_{self.produces[0]}"

    def _validate_input(self, x) -> bool:
        if not isinstance(x, pd.DataFrame):
            raise TypeError("Input must be a pandas DataFrame")
        if self.requires[0] not in x.columns:
            raise ValueError(f"Input column {self.requires[0]} not found in DataFrame")
        return True

    @staticmethod
    def _get_timestamps(x: pd.DataFrame):
        if not isinstance(x.index, pd.DatetimeIndex):
            raise ValueError("Input must have a datetime index")
        return x.index.values.astype(np.int64)

    def _prepare_input_nb(self, x: pd.DataFrame):
        return x[self.requires[0]].values

    def _prepare_output_nb(self, idx, y) -> pd.Series:
        return pd.Series(y, index=idx, name=self.output_name)

    def __call__(self, x: pd.DataFrame, *, backend: str = "nb") -> pd.Series:
        assert backend in ("pd", "nb", "hip"), "Backend must be 'pd', 'nb' or 'hip'."
        self._validate_input(x)
        return self._hip(x)

    def _hip(self, x):
        raise NotImplementedError

    def _dev(self, ts, y):
        """Device-resident form used by `Compose`: (DeviceArray timestamps, DeviceArray input) -> DeviceArray output."""
        raise NotImplementedError


class ReturnT(SISOTransform):
    """Lagged return over a time window on an irregular series (reference transforms.py:89-117)."""

    def __init__(self, window: pd.Timedelta = pd.Timedelta(seconds=1e-6), is_log: bool = False,
                 input_col: str = "close"):
        window_sec = window.total_seconds()
        super().__init__(input_col, f"ret{window_sec}s" if window_sec > 1e-6 else "ret1")
        self.window_sec = window_sec
        self.is_log = is_log

    def _hip(self, x):
        res = comp_lagged_returns(self._get_timestamps(x), self._prepare_input_nb(x), self.window_sec, self.is_log)
        return self._prepare_output_nb(x.index, res)

    def _dev(self, ts, y):
        if self.window_sec <= 0:
            raise ValueError("The return window must be greater than zero.")
        out = DeviceArray(ts.ctx, y.n, np.float64)
        ts.ctx.call("fmk_comp_lagged_returns_dev", ts.p, y.p, c_i64(y.n), c_f64(self.window_sec), C.c_int(bool(self.is_log)),
                    out.p)
        return out


class EWMST(SISOTransform):
    """Time-decay exponentially weighted std (reference transforms.py:308-332)."""

    def __init__(self, half_life: pd.Timedelta, input_col: str = "y"):
        half_life_sec = half_life.total_seconds()
        super().__init__(input_col, f"ewms{half_life_sec}s")
        self.half_life_sec = half_life_sec

    def _hip(self, x):
        res = ewmst(self._get_timestamps(x), self._prepare_input_nb(x), self.half_life_sec)
        return self._prepare_output_nb(x.index, res)

    def _dev(self, ts, y):
        out = DeviceArray(ts.ctx, y.n, np.float64)
        ts.ctx.call("fmk_ewmst_dev", ts.p, y.p, c_i64(y.n), c_f64(self.half_life_sec), c_f64(1e-12), C.c_int(0), out.p)
        return out


class RealizedVolatility(SISOTransform):
    """Rolling realised volatility of a return series (reference transforms.py:449-491)."""

    def __init__(self, window: int, is_sample: bool = False, input_col: str = "ret"):
        super().__init__(input_col, f"rv{window}")
        self.window = window
        self.is_sample = is_sample

    def _hip(self, x):
        res = realized_vol(self._prepare_input_nb(x).astype(np.float64), self.window, self.is_sample)
        return self._prepare_output_nb(x.index, res)

    def _dev(self, ts, y):
        out = DeviceArray(ts.ctx, y.n, np.float64)
        if int(self.window) == 0:                      # every window is empty (core/volatility.py realized_vol)
            return DeviceArray.from_host(ts.ctx, np.full(y.n, np.nan))
        ts.ctx.call("fmk_realized_vol_dev", y.p, c_i64(y.n), c_i64(int(self.window)), C.c_int(bool(self.is_sample)), out.p)
        return out


class Compose(SISOTransform):
    """Chain of SISO transforms; the output of step i feeds step i+1 (reference feature/kit.py Compose)."""

    def __init__(self, *transforms: SISOTransform):
        first_out = transforms[0].output_name
        super().__init__(transforms[0].requires[0], "_".join([first_out] + [t.produces[0] for t in transforms[1:]]))
        self.transforms = transforms

    @property
    def output_name(self) -> str:
        return self.produces[0]

    def __call__(self, x: pd.DataFrame, *, backend: str = "nb") -> pd.Series:
        assert backend in ("pd", "nb", "hip"), "Backend must be 'pd', 'nb' or 'hip'."
        self._validate_input(x)
        if self.output_name in x.columns:
            return x[self.output_name]
        # (only when every step has a device form of its own: a nested Compose or a user-defined SISOTransform takes the
        # generic loop below)
        if len(x) and all(type(t)._dev is not SISOTransform._dev for t in self.transforms) and \
                not any(t.produces[0] in x.columns or (i and t.requires[0] in x.columns) for i, t in enumerate(self.transforms)):
            # the chain stays in HBM: timestamps and the input column go up once, every intermediate series is a device
            # buffer handed to the next kernel, only the last one comes back (the reference round-trips pandas objects)
            ctx = _ffi.default_context()
            ts = DeviceArray.from_host(ctx, self._get_timestamps(x))
            cur = DeviceArray.from_host(ctx, np.ascontiguousarray(self.transforms[0]._prepare_input_nb(x), dtype=np.float64))
            for t in self.transforms:
                cur = t._dev(ts, cur)
            return pd.Series(cur.to_host(), index=x.index, name=self.output_name)
        cur = None
        for i, t in enumerate(self.transforms):
            if t.produces[0] in x.columns:
                cur = x[t.produces[0]]
            elif i == 0:
                cur = t(x, backend=backend)
            else:
                req = t.requires[0]
                df_in = x[[req]] if req in x.columns else pd.DataFrame(cur.values, index=cur.index, columns=[req])
                cur = t(df_in, backend=backend)
        cur.name = self.output_name
        return cur
