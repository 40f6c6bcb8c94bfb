"""ctypes binding of libfmk_hip.so (the C ABI declared in include/fmk.h).

This is the only place the Python host layer touches native code.  There is no CPU fallback:
if the shared library is missing, or no gfx950 device is usable, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfmk_hip.so")

OK = 0
E_ARG, E_CAPACITY, E_LEVEL, E_ZERODIV, E_NOMEM, E_HIP, E_NODEVICE, E_COMM = -1, -2, -3, -4, -5, -6, -7, -8

c_i64 = C.c_int64
c_f64 = C.c_double
c_vp = C.c_void_p


class DirectionalOut(C.Structure):
    _fields_ = [(k, c_vp) for k in (
        "ticks_buy", "ticks_sell", "volume_buy", "volume_sell", "dollars_buy", "dollars_sell",
        "mean_spread", "max_spread", "cum_ticks_min", "cum_ticks_max", "cum_volumes_min",
        "cum_volumes_max", "cum_dollars_min", "cum_dollars_max")]


class FootprintOut(C.Structure):
    _fields_ = [(k, c_vp) for k in (
        "price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks",
        "buy_imbalances", "sell_imbalances", "buy_imbalances_sum", "sell_imbalances_sum",
        "cot_price_levels", "imb_max_run_signed", "vp_skew", "vp_gini")]


DIRECTIONAL_FIELDS = [(k, np.int64 if "ticks" in k else np.float32) for k, _ in DirectionalOut._fields_]
FOOTPRINT_FLAT_FIELDS = [("price_levels", np.int32), ("buy_volumes", np.float32), ("sell_volumes", np.float32),
                         ("buy_ticks", np.int32), ("sell_ticks", np.int32), ("buy_imbalances", np.uint8),
                         ("sell_imbalances", np.uint8)]
FOOTPRINT_BAR_FIELDS = [("buy_imbalances_sum", np.uint16), ("sell_imbalances_sum", np.uint16),
                        ("cot_price_levels", np.int32), ("imb_max_run_signed", np.int16),
                        ("vp_skew", np.float64), ("vp_gini", np.float64)]

_lib = None
_lock = threading.Lock()


class FmkError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libfmk_hip.so (fails loudly if it was not built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise FmkError(
                        f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                        "(hipcc --offload-arch=gfx950).  finmlkit_amd has no CPU fallback.")
                l = C.CDLL(LIB_PATH)
                l.fmk_last_error.restype = C.c_char_p
                l.fmk_last_error.argtypes = [c_vp]
                l.fmk_ctx_stream.restype = c_vp
                l.fmk_ctx_stream.argtypes = [c_vp]
                if l.fmk_abi_version() != 1:
                    raise FmkError("libfmk_hip.so ABI version mismatch")
                _lib = l
    return _lib


_diag = None
DIAG_PROBES = ("fmk_diag_read_bandwidth", "fmk_diag_fill_amounts_dev", "fmk_diag_read_two_streams", "fmk_diag_hop_latency",
               "fmk_diag_h2d_rate", "fmk_diag_marker_dev", "fmk_diag_read_owned")


def diag_lib() -> C.CDLL:
    """libfmk_diag.so: the bandwidth / latency probes and the full-mantissa size generator that tests, tools and bench.py use
    (include/fmk_diag.h).  Not part of the product: nothing in finmlkit_amd calls it."""
    global _diag
    if _diag is None:
        lib()                                                    # the probes take the product library's context type
        path = os.path.join(os.path.dirname(LIB_PATH), "libfmk_diag.so")
        if not os.path.exists(path):
            path = os.path.join(_HERE, "lib", "libfmk_diag.so")
        if not os.path.exists(path):
            raise FmkError(f"{path} not found: build it with `python __graft_entry__.py build`")
        _diag = C.CDLL(path)
    return _diag


def check(rc: int, ctx=None, allow=()):
    """Map a C status to the exception type the reference raises for the same condition."""
    if rc == OK or rc in allow:
        return rc
    msg = lib().fmk_last_error(ctx).decode(errors="replace") if _lib is not None else ""
    if rc in (E_ARG, E_CAPACITY):
        raise ValueError(msg or "invalid argument")
    if rc == E_LEVEL:
        raise ValueError("Something went wrong! Invalid price level index!")
    if rc == E_ZERODIV:
        raise ZeroDivisionError("division by zero")
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise FmkError(f"libfmk_hip status {rc}: {msg}")


def ptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_vp)


def amount_array(a):
    """The amount column as the library takes it: float32 stays float32, the rest -> float64."""
    a = np.asarray(a)
    if a.dtype == np.float32:
        return np.ascontiguousarray(a), 0
    return np.ascontiguousarray(a, dtype=np.float64), 1


class Context:
    """One HIP stream on one gfx950 device (fmk_ctx).  Not re-entrant."""

    def __init__(self, device: int = 0):
        self._h = c_vp()
        self.device = device
        check(lib().fmk_ctx_create(C.c_int(device), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().fmk_ctx_destroy(self._h)
            self._h = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args, allow=()):
        l = diag_lib() if name in DIAG_PROBES else lib()
        return check(getattr(l, name)(self._h, *args), self._h, allow=allow)

    def trim(self):
        """Give back everything the context caches between calls (scratch, allocator free lists, indexer buffers)."""
        self.call("fmk_ctx_trim")

    def set_fast_threshold(self, on: bool):
        """Volume / dollar indexers: True = return the parallel result even when some decisions are uncertified (counted,
        each may differ from the reference by one tick); False (default) = such inputs are redone by the exact loop."""
        self.call("fmk_ctx_set_fast_threshold", C.c_int(1 if on else 0))

    def set_enqueue_only(self, on: bool):
        """True: calls never wait for the device where they have the choice (comp_bar_ohlcv then enqueues the launches that serve
        long bars without looking whether there are any) -- for the sharded step, which overlaps its calls with the halo exchange."""
        self.call("fmk_ctx_set_enqueue_only", C.c_int(1 if on else 0))
        self._enqueue_only = bool(on)

    @property
    def enqueue_only(self) -> bool:
        """The flag as last set through this object (False after creation)."""
        return getattr(self, "_enqueue_only", False)

    def sync(self):
        self.call("fmk_ctx_sync")

    # --- device memory -------------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = c_vp()
        self.call("fmk_alloc", C.c_size_t(int(nbytes)), C.byref(p))
        return p.value

    def free(self, dptr):
        if dptr:
            self.call("fmk_free", c_vp(dptr))

    def timer_start(self):
        self.call("fmk_timer_start")

    def timer_stop(self) -> float:
        ms = c_f64()
        self.call("fmk_timer_stop", C.byref(ms))
        return ms.value

    def mem_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        self.call("fmk_mem_info", C.byref(f), C.byref(t))
        return f.value, t.value


class DeviceArray:
    """A typed 1-D device buffer owned by a Context (freed on close()/GC)."""

    def __init__(self, ctx: Context, n: int, dtype, dptr: int | None = None, owner=None):
        self.ctx = ctx
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self._owned = dptr is None
        self._owner = owner
        self.ptr = ctx.alloc(max(1, self.n) * self.dtype.itemsize) if dptr is None else int(dptr)

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    @property
    def p(self):
        return c_vp(self.ptr)

    def view(self, start: int, count: int | None = None) -> "DeviceArray":
        count = self.n - start if count is None else count
        assert 0 <= start and start + count <= self.n
        return DeviceArray(self.ctx, count, self.dtype, self.ptr + start * self.dtype.itemsize, owner=self)

    @classmethod
    def from_host(cls, ctx: Context, a) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        d = cls(ctx, a.size, a.dtype)
        if a.size:
            ctx.call("fmk_h2d", d.p, ptr(a), C.c_size_t(a.nbytes))
        return d

    def to_host(self) -> np.ndarray:
        out = np.empty(self.n, self.dtype)
        if self.n:
            self.ctx.call("fmk_d2h", ptr(out), self.p, C.c_size_t(self.nbytes))
        return out

    def zero(self):
        self.ctx.call("fmk_memset", self.p, C.c_int(0), C.c_size_t(self.nbytes))

    def free(self):
        if self._owned and self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            if self._owned and self.ptr and self.ctx._h:
                self.ctx.free(self.ptr)
        except Exception:
            pass


def upload_columns(ctx: Context, arrays):
    """Host arrays -> new DeviceArrays by ONE fmk_h2d_columns call (worker threads, pinned staging: csrc/fmk_upload.hip)."""
    arrays = [np.ascontiguousarray(a) for a in arrays]
    dev = [DeviceArray(ctx, a.size, a.dtype) for a in arrays]
    n = len(arrays)
    if n:
        dst = (c_vp * n)(*[d.ptr for d in dev])
        src = (c_vp * n)(*[a.ctypes.data for a in arrays])
        nb = (C.c_size_t * n)(*[a.nbytes for a in arrays])
        ctx.call("fmk_h2d_columns", C.c_int(n), dst, src, nb)
    return dev


_default_ctx = None


def default_context() -> Context:
    """Process-wide context on device FMK_DEVICE (default: LOCAL_RANK or 0)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("FMK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
    return _default_ctx


def device_count() -> int:
    n = C.c_int()
    rc = lib().fmk_device_count(C.byref(n))
    return n.value if rc == OK else 0


class Event:
    """hipEvent on a context's stream (non-blocking record; elapsed after sync)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._e = c_vp()
        ctx.call("fmk_event_create", C.byref(self._e))

    def record(self):
        self.ctx.call("fmk_event_record", self._e)

    def elapsed_ms(self, stop: "Event") -> float:
        ms = c_f64()
        self.ctx.call("fmk_event_elapsed", self._e, stop._e, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if self._e and self.ctx._h:
                self.ctx.call("fmk_event_destroy", self._e)
        except Exception:
            pass
