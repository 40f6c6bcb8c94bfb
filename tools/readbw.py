#!/usr/bin/env python3
"""Achievable read-only HBM bandwidth of a plain streaming kernel (calibration of the roofline in DESIGN.md)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi
from finmlkit_amd._ffi import DeviceArray
ctx = _ffi.default_context()
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_000_000_000
buf = DeviceArray(ctx, nbytes // 8, np.int64)
buf.zero(); ctx.sync()
for variant, name in ((0, "16 B/lane"), (1, "8 B/lane"), (4, "16 B nt"), (5, "8 B nt")):
    for bpc in (4, 8, 16, 32):
        ms = C.c_double()
        best = 1e9
        for _ in range(5):
            ctx.call("fmk_diag_read_bandwidth", buf.p, C.c_size_t(nbytes), C.c_int(variant), C.c_int(bpc), C.byref(ms))
            best = min(best, ms.value)
        print(f"{name:10s} {bpc:3d} blocks/CU: {best:7.3f} ms  {nbytes / best / 1e6:8.1f} GB/s", flush=True)
