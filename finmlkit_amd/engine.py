"""Device-resident tick->bar pipeline (no PCIe traffic between stages).

`DeviceTrades` holds the four trade columns in HBM; the methods enqueue the HIP kernels of
csrc/ on the context's stream through the *_dev entry points of include/fmk.h and return
`DeviceArray`s.  The NumPy drop-in functions of finmlkit_amd.bar / finmlkit_amd.feature and
bench.py are thin layers over this module.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np

from . import _ffi
from ._ffi import (DIRECTIONAL_FIELDS, FOOTPRINT_BAR_FIELDS, FOOTPRINT_FLAT_FIELDS, Context, DeviceArray,
                   DirectionalOut, FootprintOut, c_f64, c_i64, c_vp)

DENSE_GAP_MOD = 100_000_000          # mean gap 50 ms, no empty 1-min bar
SPARSE_GAP_MOD = 500_000_000_000     # "sparse" variant: exercises empty bars

OHLCV_FIELDS = [("open", np.float64), ("high", np.float64), ("low", np.float64), ("close", np.float64),
                ("volume", np.float32), ("vwap", np.float64), ("trades", np.int64),
                ("median_trade_size", np.float64)]


class DeviceTrades:
    """Trade columns resident in HBM: ts int64 ns, price f64, amount f32|f64, side int8."""

    def __init__(self, ctx: Context, ts: DeviceArray, price: DeviceArray, amount: DeviceArray,
                 side: Optional[DeviceArray]):
        self.ctx, self.ts, self.price, self.amount, self._side = ctx, ts, price, amount, side
        self._side_host = None
        self.n = price.n
        self.amount_is_f64 = int(amount.dtype == np.float64)

    @property
    def side(self) -> Optional[DeviceArray]:
        """The side column; with from_numpy(..., lazy_side=True) it is uploaded when something first asks for it (build_ohlcv,
        the threshold indexers and the tick-level features never do)."""
        if self._side is None and self._side_host is not None:
            self._side = DeviceArray.from_host(self.ctx, self._side_host)
            self._side_host = None
        return self._side

    @side.setter
    def side(self, v):
        self._side, self._side_host = v, None

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_numpy(cls, ts, price, amount, side=None, ctx: Optional[Context] = None, lazy_side: bool = False) -> "DeviceTrades":
        ctx = ctx or _ffi.default_context()
        am, _ = _ffi.amount_array(amount)
        # np.ascontiguousarray / asarray copy only when the dtype or the layout differs (a frame's columns arrive as they are)
        host = [np.ascontiguousarray(ts, dtype=np.int64), np.ascontiguousarray(price, dtype=np.float64), am]
        sd = None if side is None else np.ascontiguousarray(side, dtype=np.int8)
        if sd is not None and not lazy_side:
            host.append(sd)
        dev = _ffi.upload_columns(ctx, host)      # ONE call for the frame (csrc/fmk_upload.hip)
        t = cls(ctx, dev[0], dev[1], dev[2], dev[3] if len(dev) > 3 else None)
        if sd is not None and lazy_side:
            t._side_host = sd                     # a reference to the caller's column (or its int8 copy); uploaded on first use
        return t

    @classmethod
    def synth(cls, n: int, seed: int = 42, first: int = 0, gap_mod: int = DENSE_GAP_MOD,
              ctx: Optional[Context] = None, headroom: int = 0, into=None) -> "DeviceTrades":
        """Ticks [first, first+n) of the synthetic stream, generated on the device.

        `headroom` reserves that many elements *in front* of every column (multi-GPU halo)."""
        ctx = ctx or _ffi.default_context()
        if into is not None:
            # the four columns carved out of ONE caller-owned allocation at byte offset `into[1]` (each column on a 2 MiB boundary):
            # bench.py's placement probe -- the level of the reducers depends on where in device memory the columns lie
            slab, off = into
            cols, at = [], int(off)
            for dt in (np.int64, np.float64, np.float32, np.int8):
                nb = (n + headroom) * np.dtype(dt).itemsize
                assert at + nb <= slab.nbytes, "DeviceTrades.synth(into=...): the slab is too small"
                cols.append(DeviceArray(ctx, n + headroom, dt, slab.ptr + at, owner=slab))
                at += (nb + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        else:
            cols = [DeviceArray(ctx, n + headroom, dt) for dt in (np.int64, np.float64, np.float32, np.int8)]
        v = [c.view(headroom, n) for c in cols]
        ctx.call("fmk_synth_trades_dev", C.c_uint64(seed), c_i64(first), c_i64(n), C.c_uint64(gap_mod),
                 v[0].p, v[1].p, v[2].p, v[3].p)
        t = cls(ctx, *v)
        t._backing = cols if into is None else []
        t._headroom = headroom
        return t

    # ------------------------------------------------------------------ placement (opt-in)
    def place(self, probe, positions: int = 7, stride: int = 16 << 30, reps: int = 6, warm: int = 3):
        """Opt-in, and it DOUBLES the device memory of the columns for the life of the returned object (the slab and the original
        columns both stay): choose WHERE in device memory the columns lie.  A trade set with head-room in front of it (a shard made for
        the halo exchange, `with_halo`) is refused -- place the columns before sharding.  The reducers that stream whole bars (k_bar_ohlcv_small ...) run up to
        10 % slower when their input columns fall into certain physical blocks of HBM (profiles/r04_placement_regions.txt: whole
        16 .. 128 GiB stretches, constant for an allocation's life; small separate allocations land in them more often than one
        large one).  `place` makes ONE allocation that holds `positions` copies' worth of address space, copies the columns to every
        `stride` bytes of it in turn, times `probe(trades)` -- the caller's own first call, e.g. `lambda t: t.time_bars_ohlcv(60.0)`
        -- on each with the context's HIP-event timer (`warm` untimed + `reps` timed calls), and returns the trade set at the fastest
        position (the original columns are left alone and also take part as position 0).

        -> (DeviceTrades, info) with info = {"probe_ms": [...], "offset_gib": [None, 0, ...], "chosen": k}.  The slab stays
        allocated for the life of the returned object (releasing it moved the level of other allocations: r04_sharded_step.txt).
        Costs positions x (a device-to-device copy of the columns + the probes): set-up, once per trade set."""
        ctx = self.ctx
        if getattr(self, "_headroom", 0):
            raise ValueError("place(): this trade set has head-room for a halo in front of its columns (a shard); the placed copy "
                             "would lose it -- place the columns before sharding")
        cols = [c for c in (self.ts, self.price, self.amount, self.side) if c is not None]
        span = sum((c.nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20) for c in cols)
        span = (span + (1 << 30) - 1) // (1 << 30) * (1 << 30)
        # positions 1 .. P-1 lie every `step` bytes of ONE slab (they may overlap: one copy lives at a time, the best is copied again
        # at the end); as many as fit beside 8 GiB of working memory
        step = max(int(stride), 2 << 20)
        free, _ = ctx.mem_info()
        k = max(0, int(positions) - 1)
        while k > 0 and (k - 1) * step + span > free - (8 << 30):
            k -= 1
        slab = None
        while k > 0 and slab is None:
            try:
                slab = DeviceArray(ctx, (k - 1) * step + span, np.uint8)
            except Exception:                                        # noqa: BLE001 -- a refused allocation: fewer positions
                k -= 1

        def timed(t):
            for _ in range(warm):
                probe(t)
            best = None
            for _ in range(reps):
                ctx.timer_start()
                probe(t)
                ms = ctx.timer_stop()
                best = ms if best is None else min(best, ms)
            return best

        def at(off):
            out, o = [], int(off)
            for c in cols:
                out.append(DeviceArray(ctx, c.n, c.dtype, slab.ptr + o, owner=slab))
                o += (c.nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
            for src, dst in zip(cols, out):
                ctx.call("fmk_d2d", dst.p, src.p, C.c_size_t(src.nbytes))
            if self.side is None:
                out.append(None)
            t = DeviceTrades(ctx, *out)
            t._first_last = getattr(self, "_first_last", None)
            return t

        ms, offs = [timed(self)], [None]
        if slab is not None:
            for i in range(k):
                ms.append(timed(at(i * step)))
                offs.append(i * step)
        best = min(range(len(ms)), key=lambda i: ms[i])
        chosen = self if best == 0 else at(offs[best])
        if chosen is not self:
            chosen._placement_keep = (slab, self)
        return chosen, {"probe_ms": ms, "offset_gib": [None if o is None else o / float(1 << 30) for o in offs], "chosen": best}

    def with_halo(self, k: int) -> "DeviceTrades":
        """View that also covers the k elements in front of the shard (already filled by the caller)."""
        assert k <= getattr(self, "_headroom", 0)
        h = self._headroom
        cols = [b.view(h - k, self.n + k) for b in self._backing]
        t = DeviceTrades(self.ctx, *cols)
        t._backing, t._headroom = self._backing, h - k
        return t

    def to_numpy(self):
        return (self.ts.to_host(), self.price.to_host(), self.amount.to_host(),
                None if self.side is None else self.side.to_host())

    # ------------------------------------------------------------------ indexers
    def first_last_ts(self) -> Tuple[int, int]:
        """timestamps[0], timestamps[-1] (what the reference reads from its host array, logic.py:33-36);
        fetched from the device once per trade set and cached -- the columns are immutable."""
        if getattr(self, "_first_last", None) is None:
            a = self.ts.view(0, 1).to_host()[0]
            b = self.ts.view(self.n - 1, 1).to_host()[0]
            self._first_last = (int(a), int(b))
        return self._first_last

    def time_bar_index(self, interval_seconds: float, clock_params=None,
                       out: Optional[Tuple[DeviceArray, DeviceArray]] = None) -> Tuple[DeviceArray, DeviceArray]:
        """_time_bar_indexer (logic.py:12-51) -> (bar_clock, bar_close_indices) on the device.

        `out` = preallocated (clock, idx) buffers with capacity >= n_edges (views of them are returned)."""
        if clock_params is None:
            t0, t1 = self.first_last_ts()
            ne, e0, d = c_i64(), c_i64(), c_i64()
            _ffi.check(_ffi.lib().fmk_time_bar_clock(c_i64(t0), c_i64(t1), c_f64(interval_seconds),
                                                     C.byref(ne), C.byref(e0), C.byref(d)))
            clock_params = (ne.value, e0.value, d.value)
        ne, e0, d = clock_params
        if out is not None:
            clock, idx = out[0].view(0, ne), out[1].view(0, ne)
        else:
            clock = DeviceArray(self.ctx, ne, np.int64)
            idx = DeviceArray(self.ctx, ne, np.int64)
        self.ctx.call("fmk_time_bar_indexer_dev", self.ts.p, c_i64(self.n), c_i64(e0), c_i64(d), c_i64(ne),
                      clock.p, idx.p)
        return clock, idx

    def time_bars_ohlcv(self, interval_seconds: float, want_median: bool = True, clock_params=None,
                        out_index: Optional[Tuple[DeviceArray, DeviceArray]] = None,
                        out: Optional[Dict[str, DeviceArray]] = None):
        """TimeBarKit.build_ohlcv on the resident columns in ONE library call (kit.py:42-66 -> base.py:126-158):
        _time_bar_indexer + comp_bar_ohlcv -> (bar_clock, bar_close_indices, ohlcv dict), the same values as
        time_bar_index() followed by bar_ohlcv().  For 1-minute-sized bars it is one kernel launch (the edge search runs
        inside the OHLCV + median kernel)."""
        if clock_params is None:
            t0, t1 = self.first_last_ts()
            ne, e0, d = c_i64(), c_i64(), c_i64()
            _ffi.check(_ffi.lib().fmk_time_bar_clock(c_i64(t0), c_i64(t1), c_f64(interval_seconds),
                                                     C.byref(ne), C.byref(e0), C.byref(d)))
            clock_params = (ne.value, e0.value, d.value)
        ne, e0, d = clock_params
        t0, t1 = self.first_last_ts()
        if out_index is not None:
            clock, idx = out_index[0].view(0, ne), out_index[1].view(0, ne)
        else:
            clock, idx = DeviceArray(self.ctx, ne, np.int64), DeviceArray(self.ctx, ne, np.int64)
        out = out or self.alloc_ohlcv(max(ne - 1, 0), want_median)
        med = out["median_trade_size"].p if (want_median and "median_trade_size" in out) else None
        self.ctx.call("fmk_time_bars_ohlcv_dev", self.ts.p, self.price.p, self.amount.p, C.c_int(self.amount_is_f64),
                      c_i64(self.n), c_i64(t0), c_i64(t1), c_i64(e0), c_i64(d), c_i64(ne), clock.p, idx.p,
                      out["open"].p, out["high"].p, out["low"].p, out["close"].p, out["volume"].p, out["vwap"].p,
                      out["trades"].p, med)
        return clock, idx, out

    def tick_bar_index(self, threshold: int) -> DeviceArray:
        m = c_i64()
        self.ctx.call("fmk_tick_bar_indexer_dev", c_i64(self.n), c_i64(int(threshold)), None, c_i64(0), C.byref(m))
        out = DeviceArray(self.ctx, m.value, np.int64)
        self.ctx.call("fmk_tick_bar_indexer_dev", c_i64(self.n), c_i64(int(threshold)), out.p, c_i64(m.value),
                      C.byref(m))
        return out

    def _threshold_index(self, fn, cols, threshold) -> DeviceArray:
        m, unc = c_i64(), c_i64()
        self.ctx.call(fn, *cols, c_i64(self.n), c_f64(threshold), None, c_i64(0), C.byref(m), C.byref(unc))
        out = DeviceArray(self.ctx, m.value, np.int64)
        self.ctx.call(fn, *cols, c_i64(self.n), c_f64(threshold), out.p, c_i64(m.value), C.byref(m), C.byref(unc))
        self.last_uncertified = unc.value
        return out

    def volume_bar_index(self, threshold: float) -> DeviceArray:
        return self._threshold_index("fmk_volume_bar_indexer_dev", (self.amount.p, C.c_int(self.amount_is_f64)),
                                     threshold)

    def dollar_bar_index(self, threshold: float) -> DeviceArray:
        return self._threshold_index("fmk_dollar_bar_indexer_dev",
                                     (self.price.p, self.amount.p, C.c_int(self.amount_is_f64)), threshold)

    def gather_ts(self, close_idx: DeviceArray) -> DeviceArray:
        out = DeviceArray(self.ctx, close_idx.n, np.int64)
        self.ctx.call("fmk_gather_i64_dev", self.ts.p, c_i64(self.n), close_idx.p, c_i64(close_idx.n), out.p)
        return out

    # ------------------------------------------------------------------ reducers
    def alloc_ohlcv(self, n_bars: int, want_median: bool = True) -> Dict[str, DeviceArray]:
        return {k: DeviceArray(self.ctx, n_bars, dt) for k, dt in OHLCV_FIELDS
                if want_median or k != "median_trade_size"}

    def bar_ohlcv(self, close_idx: DeviceArray, want_median: bool = True,
                  out: Optional[Dict[str, DeviceArray]] = None) -> Dict[str, DeviceArray]:
        """comp_bar_ohlcv (base.py:306-407) on device-resident columns."""
        nb = close_idx.n - 1
        out = out or self.alloc_ohlcv(max(nb, 0), want_median)
        med = out["median_trade_size"].p if (want_median and "median_trade_size" in out) else None
        self.ctx.call("fmk_comp_bar_ohlcv_dev", self.price.p, self.amount.p, C.c_int(self.amount_is_f64),
                      c_i64(self.n), close_idx.p, c_i64(close_idx.n), out["open"].p, out["high"].p,
                      out["low"].p, out["close"].p, out["volume"].p, out["vwap"].p, out["trades"].p, med)
        return out

    def bar_median(self, close_idx: DeviceArray, out: DeviceArray) -> DeviceArray:
        self.ctx.call("fmk_comp_bar_median_dev", self.amount.p, C.c_int(self.amount_is_f64), c_i64(self.n),
                      close_idx.p, c_i64(close_idx.n), out.p)
        return out

    def bar_trade_size(self, close_idx: DeviceArray, theta, theta_mult: float = 5.0) -> Dict[str, np.ndarray]:
        """comp_bar_trade_size_features (base.py:549-612) -> host float32 arrays keyed like the reference's
        DataFrame columns."""
        nb = close_idx.n - 1
        th = DeviceArray.from_host(self.ctx, np.ascontiguousarray(theta, dtype=np.float64))
        keys = ("mean_size_rel", "size_95_rel", "pct_block", "size_gini")
        out = {k: DeviceArray(self.ctx, nb, np.float32) for k in keys}
        self.ctx.call("fmk_comp_bar_trade_size_dev", self.amount.p, C.c_int(self.amount_is_f64), c_i64(self.n), th.p,
                      close_idx.p, c_i64(close_idx.n), c_f64(theta_mult), *[out[k].p for k in keys])
        return {k: v.to_host() for k, v in out.items()}

    def bar_directional(self, close_idx: DeviceArray) -> Tuple[Dict[str, DeviceArray], DeviceArray]:
        """comp_bar_directional_features (base.py:409-546); second value: device count of bars without a
        signed tick (the reference raises ZeroDivisionError for those)."""
        nb = close_idx.n - 1
        out = {k: DeviceArray(self.ctx, nb, dt) for k, dt in DIRECTIONAL_FIELDS}
        st = DirectionalOut(**{k: out[k].ptr for k in out})
        nz = DeviceArray(self.ctx, 1, np.int64)
        nz.zero()
        self.ctx.call("fmk_comp_bar_directional_dev", self.price.p, self.amount.p, C.c_int(self.amount_is_f64),
                      c_i64(self.n), close_idx.p, c_i64(close_idx.n), self.side.p, C.byref(st), nz.p)
        return out, nz

    def bar_footprints(self, close_idx: DeviceArray, lows: DeviceArray, highs: DeviceArray,
                       price_tick_size: float, imbalance_factor: float = 3.0):
        """comp_bar_footprints (base.py:615-850) in CSR form -> (level_offsets, flat, per_bar, n_bad)."""
        nb = close_idx.n - 1
        off = DeviceArray(self.ctx, nb + 1, np.int64)
        tot, mx = c_i64(), c_i64()
        self.ctx.call("fmk_comp_bar_footprints_size_dev", lows.p, highs.p, c_i64(nb), c_f64(price_tick_size),
                      off.p, C.byref(tot), C.byref(mx))
        flat = {k: DeviceArray(self.ctx, tot.value, dt) for k, dt in FOOTPRINT_FLAT_FIELDS}
        bar = {k: DeviceArray(self.ctx, nb, dt) for k, dt in FOOTPRINT_BAR_FIELDS}
        st = FootprintOut(**{k: v.ptr for k, v in {**flat, **bar}.items()})
        bad = DeviceArray(self.ctx, 1, np.int64)
        bad.zero()
        self.ctx.call("fmk_comp_bar_footprints_fill_dev", self.price.p, self.amount.p,
                      C.c_int(self.amount_is_f64), c_i64(self.n), close_idx.p, c_i64(close_idx.n), self.side.p,
                      c_f64(price_tick_size), lows.p, c_f64(imbalance_factor), off.p, c_i64(mx.value),
                      C.byref(st), bad.p)
        return off, flat, bar, bad

    def bars_fused(self, close_idx: DeviceArray, price_tick_size: float, imbalance_factor: float = 3.0,
                   want_median: bool = True):
        """cfg 4: build_ohlcv + build_directional_features + build_footprints semantics in two passes over the ticks
        (13 B/tick each: OHLCV + order-flow in one kernel, then the footprints, whose sweep also takes the median trade size).

        -> (ohlcv, directional, n_zero_div, level_offsets, flat, per_bar, n_bad), all device-resident."""
        nb = close_idx.n - 1
        o = self.alloc_ohlcv(max(nb, 0), want_median)
        med = o["median_trade_size"].p if want_median else None
        off = DeviceArray(self.ctx, nb + 1, np.int64)
        tot, mx = c_i64(), c_i64()
        d = {k: DeviceArray(self.ctx, nb, dt) for k, dt in DIRECTIONAL_FIELDS}
        dst = DirectionalOut(**{k: d[k].ptr for k in d})
        cnts = DeviceArray(self.ctx, 2, np.int64)
        cnts.zero()
        # pass 1: OHLCV and the order-flow features from ONE read of price / amount / side, then the level counts.  The median
        # trade size is left to pass 2 when the library says so (float32 amounts, 600..2048-tick bars): 26 B/tick in all
        deferred = C.c_int(0)
        self.ctx.call("fmk_bars_flow_size_defer_dev", self.price.p, self.amount.p, C.c_int(self.amount_is_f64), c_i64(self.n),
                      close_idx.p, c_i64(close_idx.n), self.side.p, c_f64(price_tick_size), o["open"].p, o["high"].p,
                      o["low"].p, o["close"].p, o["volume"].p, o["vwap"].p, o["trades"].p, med, C.byref(dst),
                      cnts.view(0, 1).p, off.p, C.byref(tot), C.byref(mx), C.byref(deferred))
        flat = {k: DeviceArray(self.ctx, tot.value, dt) for k, dt in FOOTPRINT_FLAT_FIELDS}
        bar = {k: DeviceArray(self.ctx, nb, dt) for k, dt in FOOTPRINT_BAR_FIELDS}
        fst = FootprintOut(**{k: v.ptr for k, v in {**flat, **bar}.items()})
        # pass 2: the footprints (+ the median of the amounts each wave has just swept)
        self.ctx.call("fmk_comp_bar_footprints_fill_median_dev", self.price.p, self.amount.p, C.c_int(self.amount_is_f64),
                      c_i64(self.n), close_idx.p, c_i64(close_idx.n), self.side.p, c_f64(price_tick_size), o["low"].p,
                      c_f64(imbalance_factor), off.p, c_i64(mx.value), C.byref(fst), cnts.view(1, 1).p,
                      med if deferred.value else None)
        return o, d, cnts.view(0, 1), off, flat, bar, cnts.view(1, 1)

    # ------------------------------------------------------------------ tick-level features
    def lagged_returns(self, window_sec: float, is_log: bool, close: Optional[DeviceArray] = None) -> DeviceArray:
        out = DeviceArray(self.ctx, self.n, np.float64)
        self.ctx.call("fmk_comp_lagged_returns_dev", self.ts.p, (close or self.price).p, c_i64(self.n),
                      c_f64(window_sec), C.c_int(bool(is_log)), out.p)
        return out

    def ewmst(self, y: DeviceArray, half_life: float, sigma_floor: float = 1e-12, mean0: bool = False) -> DeviceArray:
        out = DeviceArray(self.ctx, self.n, np.float64)
        self.ctx.call("fmk_ewmst_dev", self.ts.p, y.p, c_i64(self.n), c_f64(half_life), c_f64(sigma_floor),
                      C.c_int(bool(mean0)), out.p)
        return out

    def ewms(self, y: DeviceArray, span: int) -> DeviceArray:
        out = DeviceArray(self.ctx, y.n, np.float64)
        self.ctx.call("fmk_ewms_dev", y.p, c_i64(y.n), c_i64(int(span)), out.p)
        return out

    def realized_vol(self, r: DeviceArray, window: int, is_sample: bool) -> DeviceArray:
        out = DeviceArray(self.ctx, r.n, np.float64)
        self.ctx.call("fmk_realized_vol_dev", r.p, c_i64(r.n), c_i64(int(window)), C.c_int(bool(is_sample)), out.p)
        return out


def to_host(d: Dict[str, DeviceArray]) -> Dict[str, np.ndarray]:
    return {k: v.to_host() for k, v in d.items()}
