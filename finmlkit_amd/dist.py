"""Multi-GPU time-bar path: contiguous tick-range shards + one neighbour halo exchange.

The reference has no distributed code; this is new design.  Rank r holds ticks
[r*n, (r+1)*n) of one globally sorted stream.  Bar ids are a pure function of the timestamp and
the global clock, so only the bar that straddles a shard boundary needs stitching.  Instead of
merging partial aggregates, the left rank ships the *raw ticks* of its trailing partial bar (plus
the one tick at its last complete close, which the right rank needs as the "open edge" entry:
empty-bar price, previous side/price of the first tick) to its right neighbour, which prepends
them to its shard.  Every bar is then reduced from raw ticks by exactly the kernels of the
single-GPU path -> results identical to one GPU.  A bar is owned by the rank where it closes.

Communication: two tiny host all-gathers ONCE per trade set (first/last timestamp, halo length: the
plan is a function of the immutable columns) and, per step, ONE point-to-point send/recv per
neighbour pair (price + amount slices, ~15 KB for 1-minute bars) -- `ncclSend`/`ncclRecv` of librccl
over a single xGMI link, called from csrc/fmk_comm.hip behind the C ABI (`fmk_comm_*`), ordered
against the compute stream by events only.  No PyTorch, no all-reduce, no ring.

The planning arithmetic is plain Python integers (exact); nothing here computes bar values.
"""
from __future__ import annotations
import os

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


def sharded_tick_bar_index(offset: int, n_local: int, threshold: int):
    """Tick bars on a shard ("trivially, closed form in the global tick index", no communication): the
    GLOBAL close indices of `_tick_bar_indexer` (finmlkit/bar/logic.py:54-84) that fall into the ticks [offset, offset + n_local)
    of the stream, as an int64 array; the opening entry 0 belongs to the shard that holds tick 0.  The reference counts the
    first tick as 1 and resets to 0 at a close, so closes sit at every global index g >= 1 with (g + 1) % threshold == 0
    (every g >= 1 when threshold <= 1).  Concatenated over the shards in order this is the un-sharded result."""
    import numpy as np
    offset, n_local, threshold = int(offset), int(n_local), int(threshold)
    if n_local <= 0:                                               # an empty shard owns nothing, not even the opening entry
        return np.zeros(0, np.int64)
    lo, hi = max(offset, 1), offset + n_local                      # candidate closes: lo <= g < hi
    if threshold <= 1:
        closes = np.arange(lo, hi, dtype=np.int64)
    else:
        first = ((lo + 1 + threshold - 1) // threshold) * threshold - 1       # smallest g >= lo with (g + 1) % threshold == 0
        closes = np.arange(first, hi, threshold, dtype=np.int64)
    if offset == 0:
        closes = np.concatenate([np.zeros(1, np.int64), closes])
    return closes


class Comm:
    """One rank's handle on the node's ranks: `fmk_comm_*` of libfmk_hip.so (csrc/fmk_comm.hip) -- librccl's
    ncclSend/ncclRecv on the communicator's own HIP stream (transport "rccl"), or host-staged through the rendezvous
    segment (transport "host": tests, and with ctx=None plain host buffers).  No PyTorch anywhere.

    `path` names the rendezvous file, the same string on every rank (rank 0 creates it, it is unlinked once all ranks
    have attached).  The small all-gathers are for the SET-UP phase (host buffers, blocking); a step only calls
    exchange() / wait(), which enqueue and return."""

    def __init__(self, ctx, rank: int, world: int, path: str, transport: str = "rccl", self_loop: bool = False,
                 ring_bytes: int = 0, timeout_s: float = 120.0):
        import ctypes as C
        from . import _ffi
        self._C, self._ffi = C, _ffi
        self.ctx, self.rank, self.world, self.transport = ctx, int(rank), int(world), transport
        self._h = C.c_void_p()
        kind = {"rccl": 0, "host": 1}[transport]
        lib = _ffi.lib()
        lib.fmk_comm_last_error.restype = C.c_char_p
        lib.fmk_comm_last_error.argtypes = [C.c_void_p]
        rc = lib.fmk_comm_create(ctx.handle if ctx is not None else None, C.c_int(kind), path.encode(), C.c_int(rank),
                                 C.c_int(world), C.c_int(1 if self_loop else 0), C.c_size_t(ring_bytes),
                                 C.c_double(timeout_s), C.byref(self._h))
        _ffi.check(rc, ctx.handle if ctx is not None else None)

    def _call(self, name, *args):
        rc = getattr(self._ffi.lib(), name)(self._h, *args)
        if rc != 0:
            msg = self._ffi.lib().fmk_comm_last_error(self._h).decode(errors="replace")
            if rc == self._ffi.E_ARG:
                raise ValueError(msg)
            raise self._ffi.FmkError(f"{name}: status {rc}: {msg}")

    def _gather(self, vals, np_dtype):
        import numpy as np
        a = np.ascontiguousarray(vals, dtype=np_dtype)
        out = np.empty((self.world, a.size), dtype=np_dtype)
        self._call("fmk_comm_allgather", self._ffi.ptr(a), self._C.c_size_t(a.nbytes), self._ffi.ptr(out))
        return out

    def all_gather_i64(self, vals: Sequence[int]) -> List[List[int]]:
        import numpy as np
        return [[int(x) for x in row] for row in self._gather(list(vals), np.int64)]

    def all_gather_f64(self, vals: Sequence[float]) -> List[List[float]]:
        import numpy as np
        return [[float(x) for x in row] for row in self._gather(list(vals), np.float64)]

    def barrier(self):
        """Host barrier over the ranks (callers synchronise their own stream first)."""
        self._call("fmk_comm_barrier")

    def exchange(self, send: Sequence[Tuple[int, int]], recv: Sequence[Tuple[int, int]]):
        """(pointer, bytes) column slices: `send` to rank+1, `recv` from rank-1 -- one ncclGroup (or one host-staged
        round).  Both lists must have one entry per column; a side without a neighbour passes zero lengths."""
        C = self._C
        n = max(len(send), len(recv))
        pad = lambda xs: list(xs) + [(0, 0)] * (n - len(xs))
        send, recv = pad(send), pad(recv)
        sp = (C.c_void_p * n)(*[p for p, _ in send])
        sb = (C.c_size_t * n)(*[b for _, b in send])
        rp = (C.c_void_p * n)(*[p for p, _ in recv])
        rb = (C.c_size_t * n)(*[b for _, b in recv])
        self._call("fmk_comm_halo_exchange_dev", C.c_int(n), sp, sb, rp, rb)

    def wait(self):
        """Make the context's stream wait (an event, not the host) for the exchange enqueued last."""
        self._call("fmk_comm_wait_dev")

    def sync(self):
        """Host wait for the communicator's stream, bounded by the communicator's timeout (FmkError when a peer is gone)."""
        self._call("fmk_comm_sync")

    def profile_enable(self, on: bool = True):
        """Time the next (up to 64) exchanges on their own: RCCL = event pair around the ncclGroup, host = wall clock."""
        self._call("fmk_comm_profile_enable", self._C.c_int(1 if on else 0))

    def profile_read(self) -> List[float]:
        C = self._C
        ms, n = (C.c_double * 64)(), C.c_int()
        self._call("fmk_comm_profile_read", ms, C.c_int(64), C.byref(n))
        return [ms[i] for i in range(n.value)]

    def close(self):
        if self._h:
            self._ffi.lib().fmk_comm_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedTimeBars:
    """One rank's side of a sharded `TimeBarKit.build_ohlcv()` step.

    SET-UP (host, once per trade set and interval -- every quantity is a function of the immutable columns, like
    `DeviceTrades.first_last_ts`):

      span()                 -> (first, last) timestamp of the shard                     [all-gather #1]
      make_plan(all_span)    -> halo length this rank sends right                        [all-gather #2]
                                (global clock, edge plan, local close indices of the first and last edge)
      set_halo(recv_h)       -> boundary buffers sized from the plan: [halo | head of the shard up to its first close]

    STEP (device only: nothing below reads back, waits or allocates):

      send_slices() / recv_slices()  the (pointer, bytes) lists of the one neighbour exchange
      enqueue_interior()     local close indices + every bar that needs no halo (all of them on rank 0) -- enqueued
                             while the halo travels on the communicator's stream
      enqueue_boundary()     the bar straddling the left boundary, reduced from [halo | head] by the same kernels

    `setup(comm)` / `step(comm)` run the phases over a `Comm`; tests/test_gpu_dist.py drives them with virtual ranks
    and a device copy.  The raw ticks travel, not partial aggregates: every bar is reduced by the single-GPU kernels
    from the same ticks in the same order, so the outputs are bit-identical to an un-sharded run.
    """

    def __init__(self, trades, rank: int, world: int, interval_seconds: float, want_median: bool = True,
                 with_side: bool = False, self_loop: bool = False):
        import numpy as np
        from ._ffi import DeviceArray
        self.t, self.rank, self.world = trades, rank, world
        self.interval, self.want_median = float(interval_seconds), want_median
        self.with_side = with_side and trades.side is not None
        self.self_loop = self_loop and world == 1
        self.ctx = trades.ctx
        self._np, self._DA = np, DeviceArray
        self.clock = self.idx = self.out = None
        self.plan = None
        self.send_start = trades.n
        self.recv_h = 0
        self._bnd = None

    # ------------------------------------------------------------------ set-up
    def span(self) -> Tuple[int, int]:
        return self.t.first_last_ts()

    def _cols(self, t):
        cols = [t.price, t.amount]
        if self.with_side:
            cols.append(t.side)
        return cols

    def make_plan(self, all_span: Sequence[Sequence[int]]) -> int:
        import ctypes as C
        from . import _ffi
        from ._ffi import c_f64, c_i64
        np = self._np
        ne, e0, d = c_i64(), c_i64(), c_i64()
        _ffi.check(_ffi.lib().fmk_time_bar_clock(c_i64(int(all_span[0][0])), c_i64(int(all_span[-1][1])),
                                                 c_f64(self.interval), C.byref(ne), C.byref(e0), C.byref(d)))
        self.gclock = (ne.value, e0.value, d.value)
        self.plan = plan_edges([int(a[0]) for a in all_span], *self.gclock)[self.rank]
        my = self.plan
        self.n_edges = my.hi - my.lo + 1
        self.e_lo = e0.value + my.lo * d.value
        cap = self.n_edges + 1                                  # one spare output slot (self-loop diagnostic)
        self._clock = self._DA(self.ctx, cap, np.int64)
        self._idx = self._DA(self.ctx, cap, np.int64)
        self._out = self.t.alloc_ohlcv(cap, self.want_median)
        self.out = {k: v.view(0, self.n_edges - 1) for k, v in self._out.items()}
        # close indices of my edges in LOCAL coordinates (entry 0 is -1 on ranks > 0: the open edge lies left)
        self._index()
        self.c_first = int(self.idx.view(1, 1).to_host()[0])     # read back ONCE, here
        c_last = int(self.idx.view(self.n_edges - 1, 1).to_host()[0])
        last = self.rank + 1 == self.world and not self.self_loop
        self.send_start = self.t.n if last else c_last
        return self.t.n - self.send_start

    def set_halo(self, recv_h: int):
        np = self._np
        self.recv_h = int(recv_h)
        if self.rank == 0 and not self.self_loop:
            if recv_h:
                raise RuntimeError("rank 0 has no left neighbour")
            return
        if recv_h < 1:
            raise RuntimeError(f"rank {self.rank}: empty halo")
        head = self.c_first + 1                                  # my ticks up to the first close
        self._head = head
        self._bnd = [self._DA(self.ctx, recv_h + head, c.dtype) for c in self._cols(self.t)]
        # [open edge, first close] in [halo | head] coordinates: the halo's first tick is the open edge
        self._ci0 = self._DA.from_host(self.ctx, np.array([0, recv_h + head - 1], dtype=np.int64))
        from .engine import DeviceTrades
        self._bt = DeviceTrades(self.ctx, None, self._bnd[0], self._bnd[1], self._bnd[2] if self.with_side else None)
        slot = self.n_edges - 1 if self.self_loop else 0          # the diagnostic's extra bar goes to the spare slot
        self._bout = {k: v.view(slot, 1) for k, v in self._out.items()}

    def setup(self, comm: "Comm"):
        """Both all-gathers of the set-up phase over `comm`; afterwards step(comm) can run any number of times."""
        send_h = self.make_plan(comm.all_gather_i64(list(self.span())))
        allh = comm.all_gather_i64([send_h])
        self.set_halo(allh[self.rank - 1][0] if self.rank > 0 else (send_h if self.self_loop else 0))
        return self

    # ------------------------------------------------------------------ step
    def send_slices(self) -> List[Tuple[int, int]]:
        k = self.t.n - self.send_start
        return [(c.ptr + self.send_start * c.dtype.itemsize, k * c.dtype.itemsize) for c in self._cols(self.t)]

    def recv_slices(self) -> List[Tuple[int, int]]:
        if self._bnd is None:
            return [(0, 0) for _ in self._cols(self.t)]
        return [(b.ptr, self.recv_h * b.dtype.itemsize) for b in self._bnd]

    def _index(self):
        self.clock, self.idx = self.t.time_bar_index(self.interval, clock_params=(self.n_edges, self.e_lo, self.gclock[2]),
                                                     out=(self._clock, self._idx))

    def enqueue_interior(self):
        # The shard's clock edges and its bars in ONE pipelined call (fmk_time_bars_ohlcv_dev, round 4: index stage 1 -> OHLCV launch 1 ||
        # index stage 2 -> OHLCV launch 2; the long-bar census comes from the index stages, so nothing waits for a kernel).  Bars that
        # need no halo: all of them on rank 0, bars 1.. elsewhere.  On the other ranks bar 0 is computed here from the local ticks alone
        # -- a partial bar -- and OVERWRITTEN by the boundary launch that follows on the same stream (enqueue_boundary).
        if self.n_edges > 2 or self.rank == 0:
            self.clock, self.idx, _ = self.t.time_bars_ohlcv(self.interval, self.want_median,
                                                             clock_params=(self.n_edges, self.e_lo, self.gclock[2]),
                                                             out_index=(self._clock, self._idx), out=self.out)
        else:
            self._index()

    def enqueue_boundary(self) -> int:
        import ctypes as C
        if self._bnd is not None:
            cols = self._cols(self.t)
            n = len(cols)
            src = (C.c_void_p * n)(*[c.ptr for c in cols])
            dst = (C.c_void_p * n)(*[b.ptr + self.recv_h * b.dtype.itemsize for b in self._bnd])
            nb = (C.c_size_t * n)(*[self._head * c.dtype.itemsize for c in cols])
            self.ctx.call("fmk_copy_cols_dev", C.c_int(n), src, dst, nb)       # head of the shard behind the halo
            self._bt.bar_ohlcv(self._ci0, want_median=self.want_median, out=self._bout)
        return self.plan.n_bars

    def step(self, comm: "Comm") -> int:
        # enqueue only: comp_bar_ohlcv must not wait for its first kernel here (it would, to skip the launches that serve long bars
        # when there are none) -- two host waits per step cost 0.9 ms against the 0.25 ms of launches they save
        # (the caller's setting is put back afterwards: a pipelined loop that had the flag on keeps it)
        was = self.ctx.enqueue_only
        self.ctx.set_enqueue_only(True)
        try:
            comm.exchange(self.send_slices(), self.recv_slices())    # enqueued on the communicator's stream
            self.enqueue_interior()                                   # overlaps the exchange
            comm.wait()                                               # event: context stream after the exchange
            return self.enqueue_boundary()
        finally:
            self.ctx.set_enqueue_only(was)

    def features(self, price_tick_size: float, imbalance_factor: float = 3.0):
        """cfg 4 on the shard (after a step with with_side=True): order-flow + footprints of this rank's bars through
        the same kernels as one GPU -- interior bars in local coordinates, the boundary bar from [halo | head].

        -> (directional dict, level_counts int64[B], flat dict, per-bar dict) as host arrays in bar order.  The
        spread columns of the global stream's very first bar use the reference's wrap-around tick prices[-1]
        (base.py:485-500), which on rank 0 is the last tick of the SHARD: undefined across shards, as in the
        single-GPU path it is "the last tick of whatever array is passed"."""
        np = self._np
        from .engine import to_host
        parts = []
        if self.rank > 0:
            parts.append(self._bt.bars_fused(self._ci0, price_tick_size, imbalance_factor, want_median=False))
            if self.n_edges > 2:
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


# ------------------------------------------------------------------------------------------------------------------------
# First contact with a node: `python -m finmlkit_amd.dist --selftest` (one process per rank, RANK / WORLD_SIZE / LOCAL_RANK /
# MASTER_PORT from the launcher as for bench.py; without them: one rank, RCCL self-loop).  bench.py calls selftest() at the top of
# every N > 1 run, so that gpurun_out/bench_rank<r>.log says what the node looked like BEFORE anything was timed.
# ------------------------------------------------------------------------------------------------------------------------
def describe_node() -> str:
    """fmk_comm_describe (include/fmk.h): devices, peer-access matrix, librccl path + version, the IPC environment."""
    import ctypes as C
    from . import _ffi
    buf = C.create_string_buffer(16384)
    _ffi.lib().fmk_comm_describe(buf, C.c_size_t(len(buf)))
    return buf.value.decode(errors="replace")


def selftest(rank: int, world: int, path: str, ctx=None, log=None, deadline_s: float = 10.0, payload: int = 1024) -> dict:
    """What a first N > 1 run needs to know, in this order: the node as this process sees it; a communicator over RCCL (deadline
    `deadline_s`); ONE exchange of `payload` bytes to rank + 1 (self-loop when world == 1) whose content is checked on arrival and
    whose device time is read back.  A failure of the RCCL leg is reported with librccl's own message and the same exchange is
    repeated over the host-staged transport, so that the report separates "the flow is broken" from "RCCL is broken".
    -> {"rccl": True/False, "rccl_error": str|None, "rccl_ms": float|None, "host": True/False/None, "report": str}"""
    import numpy as np
    from . import _ffi
    from ._ffi import DeviceArray
    say = (lambda m: print(m, file=log, flush=True)) if log is not None else (lambda m: None)
    res = {"rccl": False, "rccl_error": None, "rccl_ms": None, "host": None}
    report = describe_node()
    say(f"[selftest] rank {rank} of {world}: pid {os.getpid()}, LOCAL_RANK {os.environ.get('LOCAL_RANK')}, FMK_DEVICE {os.environ.get('FMK_DEVICE')}")
    for line in report.rstrip().splitlines():
        say("[selftest]   " + line)
    ctx = ctx or _ffi.default_context()
    n = payload // 8
    pattern = (np.arange(n, dtype=np.int64) * 2654435761 + (rank + 1) * 1_000_003)
    want_from = (rank - 1) % world if world > 1 else rank
    expect = (np.arange(n, dtype=np.int64) * 2654435761 + (want_from + 1) * 1_000_003)

    def one(transport: str, p: str):
        comm = Comm(ctx, rank, world, p, transport, self_loop=(world == 1), timeout_s=deadline_s)
        try:
            send = DeviceArray.from_host(ctx, pattern)
            recv = DeviceArray(ctx, n, np.int64)
            recv.zero()
            ctx.sync()
            has_right, has_left = (world == 1 or rank + 1 < world), (world == 1 or rank > 0)
            comm.profile_enable(True)
            comm.exchange([(send.ptr, payload if has_right else 0)], [(recv.ptr, payload if has_left else 0)])
            comm.wait()
            comm.sync()
            ctx.sync()
            ms = comm.profile_read()
            ok = (not has_left) or bool(np.array_equal(recv.to_host(), expect))
            oks = comm.all_gather_i64([1 if ok else 0])
            return all(r[0] == 1 for r in oks), (ms[-1] if ms else None)
        finally:
            comm.close()

    try:
        ok, ms = one("rccl", path)
        res["rccl"], res["rccl_ms"] = ok, ms
        say(f"[selftest] rank {rank}: RCCL exchange of {payload} B to rank + 1: {'ok' if ok else 'PAYLOAD MISMATCH on some rank'}"
            + (f", {ms * 1e3:.1f} us on the communicator's stream" if ms is not None else ""))
    except Exception as e:                                                     # noqa: BLE001 -- whatever librccl / the rendezvous said
        res["rccl_error"] = f"{type(e).__name__}: {e}"
        say(f"[selftest] rank {rank}: RCCL leg FAILED within its {deadline_s:.0f} s deadline: {res['rccl_error']}")
    if not res["rccl"]:
        try:
            ok, ms = one("host", path + ".selftest_host")
            res["host"] = ok
            say(f"[selftest] rank {rank}: the same exchange host-staged: {'ok' if ok else 'PAYLOAD MISMATCH'} -- the flow works, RCCL does not"
                if ok else f"[selftest] rank {rank}: the same exchange host-staged: PAYLOAD MISMATCH")
        except Exception as e:                                                 # noqa: BLE001
            res["host"] = False
            say(f"[selftest] rank {rank}: host-staged leg FAILED too: {type(e).__name__}: {e}")
    res["report"] = report
    return res


def _selftest_main(argv) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog="python -m finmlkit_amd.dist", description="first-contact self test of the multi-GPU path")
    ap.add_argument("--selftest", action="store_true", required=True)
    ap.add_argument("--deadline", type=float, default=10.0, help="seconds the RCCL leg may take")
    ap.add_argument("--one-device", action="store_true", help="every rank on device 0 (boxes with one GPU: RCCL refuses, the report says so)")
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("FMK_DEVICE", "0" if a.one_device else os.environ.get("LOCAL_RANK", "0"))
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.environ.get("FMK_BENCH_RDV") or os.path.join(base, f"fmk_selftest_{os.environ.get('MASTER_PORT', '0')}_{os.getppid() if world > 1 else os.getpid()}")
    import sys
    r = selftest(rank, world, path, log=sys.stdout, deadline_s=a.deadline)
    print(f"SELFTEST rank {rank}/{world}: " + ("RCCL ok" if r["rccl"] else f"RCCL FAILED ({r['rccl_error']}); host-staged {'ok' if r['host'] else 'FAILED'}"), flush=True)
    return 0 if r["rccl"] else 3


if __name__ == "__main__":
    import sys
    sys.exit(_selftest_main(sys.argv[1:]))
