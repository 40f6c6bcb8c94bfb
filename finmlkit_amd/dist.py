"""Multi-GPU time-bar path: contiguous tick-range shards + one neighbour halo exchange.

The reference has no distributed code; this is new design.  Rank r holds ticks
[r*n, (r+1)*n) of one globally sorted stream.  Bar ids are a pure function of the timestamp and
the global clock, so only the bar that straddles a shard boundary needs stitching.  Instead of
merging partial aggregates, the left rank ships the *raw ticks* of its trailing partial bar (plus
the one tick at its last complete close, which the right rank needs as the "open edge" entry:
empty-bar price, previous side/price of the first tick) to its right neighbour, which prepends
them to its shard.  Every bar is then reduced from raw ticks by exactly the kernels of the
single-GPU path -> results identical to one GPU.  A bar is owned by the rank where it closes.

Communication per step: two tiny all-gathers (first/last timestamp, halo length) and ONE
point-to-point send/recv per neighbour pair (4 column slices, ~25 KB for 1-minute bars) --
`torch.distributed` P2P, i.e. RCCL send/recv over a single xGMI link on the GPU box and gloo in
the CPU tests.  No all-reduce, no ring.

The planning arithmetic is plain Python integers (exact); nothing here computes bar values.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple


@dataclass
class EdgePlan:
    """Edge indices [lo, hi] of the global clock handled by one rank (bars lo..hi-1)."""
    lo: int
    hi: int

    @property
    def n_bars(self) -> int:
        return self.hi - self.lo


def plan_edges(first_ts: Sequence[int], n_edges: int, e0: int, d: int) -> List[EdgePlan]:
    """Partition the global clock edges e_k = e0 + k*d among ranks by their first timestamps.

    Edge k closes on rank r iff F_r <= e_k < F_{r+1} (F_0 = -inf, F_W = +inf): every tick <= e_k
    then lives on ranks <= r.  Rank r additionally takes the last edge before F_r as its open edge.
    """
    W = len(first_ts)
    if n_edges < 2 or d <= 0:
        raise ValueError("need at least two clock edges")

    def first_edge_ge(t: int) -> int:          # smallest k with e0 + k*d >= t, clamped to [0, n_edges]
        k = -((e0 - t) // d)                    # ceil((t - e0) / d)
        return max(0, min(n_edges, k))

    m = [0] + [first_edge_ge(int(first_ts[r])) for r in range(1, W)] + [n_edges]
    plans = []
    for r in range(W):
        lo = 0 if r == 0 else m[r] - 1
        hi = n_edges - 1 if r == W - 1 else m[r + 1] - 1
        if hi <= lo or lo < 0:
            raise ValueError(f"rank {r}: shard does not contain a complete bar close "
                             f"(edges {lo}..{hi}); use fewer ranks or shorter bars")
        plans.append(EdgePlan(lo, hi))
    return plans


class Comm:
    """Minimal wrapper over torch.distributed (nccl == RCCL on ROCm, gloo on CPU)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else torch.device("cpu")

    def all_gather_i64(self, vals: Sequence[int]) -> List[List[int]]:
        t = self.torch.tensor(list(vals), dtype=self.torch.int64, device=self.device)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[int(x) for x in o.cpu().tolist()] for o in out]

    def barrier(self):
        if self.device.type == "cuda":
            self.dist.barrier(device_ids=[self.device.index])
        else:
            self.dist.barrier()

    def neighbour_exchange(self, send_right: Sequence, recv_left: Sequence):
        """Send `send_right` tensors to rank+1 and receive `recv_left` tensors from rank-1 (one batch)."""
        ops = []
        if self.rank + 1 < self.world:
            ops += [self.dist.P2POp(self.dist.isend, t, self.rank + 1) for t in send_right]
        if self.rank > 0:
            ops += [self.dist.P2POp(self.dist.irecv, t, self.rank - 1) for t in recv_left]
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()


class ShardedTimeBars:
    """One rank's side of a sharded `TimeBarKit.build_ohlcv()` step, split into phases.

    The exchange between the phases is the caller's: RCCL in bench.py, an in-process device copy between
    virtual ranks in tests/test_gpu_dist.py.  Phases of one step:

      span()                 -> (first, last) timestamp of the shard            [all-gather #1]
      launch_local(all_span) -> halo length this rank sends right               [all-gather #2, halo send/recv]
                                (global clock + edge plan, local close indices, and -- already enqueued on the
                                context's stream, overlapping the exchange -- every bar that needs no halo)
      finish(recv_h)         -> number of bars; the bar straddling the left boundary is reduced from
                                [halo | shard] by the same kernels.

    `trades` must have been created with headroom (engine.DeviceTrades.synth(..., headroom=H)); the halo lands in
    backing[H - recv_h : H], the halo sent right is backing[H + send_start : H + n].
    """

    def __init__(self, trades, rank: int, world: int, interval_seconds: float, want_median: bool = True):
        import numpy as np
        from ._ffi import DeviceArray
        self.t, self.rank, self.world = trades, rank, world
        self.interval, self.want_median = float(interval_seconds), want_median
        self.ctx = trades.ctx
        self.headroom = trades._headroom
        self._np, self._DA = np, DeviceArray
        self._cap = 0
        self._ci0 = DeviceArray(self.ctx, 2, np.int64)
        self.clock = self.idx = self.out = None
        self.plan = None
        self.send_start = trades.n

    def span(self) -> Tuple[int, int]:
        return self.t.first_last_ts()

    def _ensure(self, ne: int):
        if self._cap < ne:
            self._cap = ne + 1024
            self._clock = self._DA(self.ctx, self._cap, self._np.int64)
            self._idx = self._DA(self.ctx, self._cap, self._np.int64)
            self._out = self.t.alloc_ohlcv(self._cap, self.want_median)

    def launch_local(self, all_span: Sequence[Sequence[int]]) -> int:
        import ctypes as C
        from . import _ffi
        from ._ffi import c_f64, c_i64
        ne, e0, d = c_i64(), c_i64(), c_i64()
        _ffi.check(_ffi.lib().fmk_time_bar_clock(c_i64(int(all_span[0][0])), c_i64(int(all_span[-1][1])),
                                                 c_f64(self.interval), C.byref(ne), C.byref(e0), C.byref(d)))
        self.gclock = (ne.value, e0.value, d.value)
        self.plan = plan_edges([int(a[0]) for a in all_span], *self.gclock)[self.rank]
        my = self.plan
        n_edges = my.hi - my.lo + 1
        self._ensure(n_edges)
        self.e_lo = e0.value + my.lo * d.value
        # close indices of my edges in LOCAL coordinates (entry 0 is -1 on ranks > 0: the open edge lies left)
        self.clock, self.idx = self.t.time_bar_index(self.interval, clock_params=(n_edges, self.e_lo, d.value),
                                                     out=(self._clock, self._idx))
        self.out = {k: v.view(0, n_edges - 1) for k, v in self._out.items()}
        last = self.rank + 1 == self.world
        # one 8-byte read-back; it also orders this step's halo receive after the previous step's kernels
        c_last = int(self.idx.view(n_edges - 1, 1).to_host()[0])
        if last:
            c_last = self.t.n
        self.send_start = c_last
        # bars that need no halo: all of them on rank 0, bars 1.. elsewhere (their ticks are local and the
        # halo only prepends, so local coordinates are valid)
        if self.rank == 0:
            self.t.bar_ohlcv(self.idx, want_median=self.want_median, out=self.out)
        elif n_edges > 2:
            self.t.bar_ohlcv(self.idx.view(1), want_median=self.want_median,
                             out={k: v.view(1) for k, v in self.out.items()})
        return 0 if last else self.t.n - c_last

    def finish(self, recv_h: int) -> int:
        from ._ffi import c_i64
        if self.rank > 0:
            if recv_h < 1 or recv_h > self.headroom:
                raise RuntimeError(f"rank {self.rank}: halo of {recv_h} ticks (headroom {self.headroom})")
            th = self.t.with_halo(recv_h)
            # [open edge, first close] in [halo | shard] coordinates: the open edge is the halo's first tick
            self.ctx.call("fmk_time_bar_indexer_dev", th.ts.p, c_i64(th.n), c_i64(self.e_lo), c_i64(self.gclock[2]),
                          c_i64(2), None, self._ci0.p)
            th.bar_ohlcv(self._ci0, want_median=self.want_median, out={k: v.view(0, 1) for k, v in self.out.items()})
        return self.plan.n_bars


    def features(self, recv_h: int, price_tick_size: float, imbalance_factor: float = 3.0):
        """cfg 4 on the shard (after `finish`): order-flow + footprints of this rank's bars through the same
        kernels as one GPU -- interior bars in local coordinates, the boundary bar from [halo | shard].

        -> (directional dict, level_counts int64[B], flat dict, per-bar dict) as host arrays in bar order.  The
        spread columns of the global stream's very first bar use the reference's wrap-around tick prices[-1]
        (base.py:485-500), which on rank 0 is the last tick of the SHARD: undefined across shards, as in the
        single-GPU path it is "the last tick of whatever array is passed"."""
        np = self._np
        from .engine import to_host
        parts = []
        n_edges = self.plan.n_bars + 1
        if self.rank > 0:
            th = self.t.with_halo(recv_h)
            parts.append(th.bars_fused(self._ci0, price_tick_size, imbalance_factor, want_median=False))
            if n_edges > 2:
                parts.append(self.t.bars_fused(self.idx.view(1), price_tick_size, imbalance_factor, want_median=False))
        else:
            parts.append(self.t.bars_fused(self.idx, price_tick_size, imbalance_factor, want_median=False))
        ds, lv, fl, pb = [], [], [], []
        for o, d, nz, off, flat, bar, bad in parts:
            if int(bad.to_host()[0]):
                raise ValueError("Something went wrong! Invalid price level index!")
            ds.append(to_host(d))
            lv.append(np.diff(off.to_host()))
            fl.append(to_host(flat))
            pb.append(to_host(bar))
        cat = lambda dicts: {k: np.concatenate([x[k] for x in dicts]) for k in dicts[0]}
        return cat(ds), np.concatenate(lv), cat(fl), cat(pb)


class ShardedTickLevel:
    """Tick-level volatility loops on a shard of one stream: `comp_lagged_returns` needs the left
    neighbour's ticks of the last `window` seconds (raw-tick halo, same mechanism as the bars), `ewmst` is a scan over
    affine maps, so a shard needs (a) its left neighbour's last tick for the first time step and (b) the state the
    earlier shards leave behind = their maps composed in shard order (one all-gather of 6 doubles per rank).

    Phases, exchange supplied by the caller (RCCL / gloo / in-process copy in tests):
      returns_send_start(next_first_ts, window) -> first local tick the RIGHT neighbour needs        [halo send/recv]
      lagged_returns(recv_h, window, is_log)     -> returns of [halo | shard] (device, h + n values; local part = [h:])
      set_left_value(r_ext, recv_h, y_left)      -> plant the left neighbour's LAST return in front of the local part
      ewmst_map(r_ext, recv_h, half_life)        -> this shard's map (6 doubles)                      [all-gather]
      ewmst(r_ext, recv_h, maps_of_lower_ranks, half_life) -> local sigma (device, n values)
    """

    def __init__(self, trades, rank: int, world: int):
        import numpy as np
        from ._ffi import DeviceArray
        self.t, self.rank, self.world, self.ctx = trades, rank, world, trades.ctx
        self._np, self._DA = np, DeviceArray

    def returns_send_start(self, next_first_ts: int, window_sec: float) -> int:
        """Index of the first local tick the right neighbour needs: one tick before `next_first_ts - window`."""
        from ._ffi import c_i64
        one = self._DA(self.ctx, 1, self._np.int64)
        edge = int(next_first_ts) - int(window_sec * 1e9) - 1024            # float64 timestamps: 256 ns granularity
        self.ctx.call("fmk_time_bar_indexer_dev", self.t.ts.p, c_i64(self.t.n), c_i64(edge), c_i64(1), c_i64(1), None,
                      one.p)
        idx = int(one.to_host()[0])                                          # last tick <= edge
        if idx < 0:
            raise ValueError(f"rank {self.rank}: the return window is longer than the shard; use fewer ranks")
        return idx

    def lagged_returns(self, recv_h: int, window_sec: float, is_log: bool):
        th = self.t.with_halo(recv_h) if recv_h else self.t
        return th.lagged_returns(window_sec, is_log)

    def set_left_value(self, r_ext, recv_h: int, y_left: float):
        import ctypes as C
        v = self._np.array([y_left], dtype=self._np.float64)
        self.ctx.call("fmk_h2d", r_ext.view(recv_h - 1, 1).p, v.ctypes.data_as(C.c_void_p), C.c_size_t(8))

    def _ext(self, r_ext, recv_h: int):
        if self.rank == 0:
            return self.t.ts, r_ext
        return self.t.with_halo(1).ts, r_ext.view(recv_h - 1, self.t.n + 1)

    def ewmst_map(self, r_ext, recv_h: int, half_life: float, mean0: bool = False):
        import ctypes as C
        from ._ffi import c_f64, c_i64
        ts, y = self._ext(r_ext, recv_h)
        m = self._DA(self.ctx, 6, self._np.float64)
        self.ctx.call("fmk_ewmst_shard_map_dev", ts.p, y.p, c_i64(y.n), c_f64(half_life), C.c_int(bool(mean0)), m.p)
        return m.to_host()

    @staticmethod
    def incoming_state(maps_of_lower_ranks):
        """(V, V2, Sy, Syy) after the lower ranks: their maps x -> a*x + b applied in order to the zero state."""
        V = V2 = Sy = Syy = 0.0
        for a, a2, bV, bV2, bSy, bSyy in maps_of_lower_ranks:
            V, V2, Sy, Syy = a * V + bV, a2 * V2 + bV2, a * Sy + bSy, a * Syy + bSyy
        return V, V2, Sy, Syy

    def ewmst(self, r_ext, recv_h: int, maps_of_lower_ranks, half_life: float, sigma_floor: float = 1e-12,
              mean0: bool = False):
        import ctypes as C
        from ._ffi import c_f64, c_i64
        ts, y = self._ext(r_ext, recv_h)
        out = self._DA(self.ctx, y.n, self._np.float64)
        st = None
        if self.rank > 0:
            st = self._DA.from_host(self.ctx, self._np.array(self.incoming_state(maps_of_lower_ranks), dtype=self._np.float64))
        self.ctx.call("fmk_ewmst_shard_apply_dev", ts.p, y.p, c_i64(y.n), c_f64(half_life), c_f64(sigma_floor),
                      C.c_int(bool(mean0)), None if st is None else st.p, out.p)
        self.ctx.sync()                                                       # `st` must outlive the kernel
        return out if self.rank == 0 else out.view(1, self.t.n)


def halo_lengths(comm: Comm, n_local: int, close_of_last_edge: int) -> Tuple[int, int]:
    """(halo I send, halo I receive).  The halo is ticks [close_of_last_edge, n_local)."""
    send = n_local - close_of_last_edge if comm.rank + 1 < comm.world else 0
    allh = comm.all_gather_i64([send])
    recv = allh[comm.rank - 1][0] if comm.rank > 0 else 0
    return send, recv
