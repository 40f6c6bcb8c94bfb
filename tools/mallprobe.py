#!/usr/bin/env python3
"""Does a re-read of a buffer that fits the 256 MiB Infinity Cache run faster than a stream from HBM?  fmk_diag_read_bandwidth (the
streaming read probe of libfmk_diag) over buffers of 16 MB .. 8 GB, six passes back to back: the first pass comes from HBM, the later
ones from wherever the lines stayed.  Also two DIFFERENT kernels' worth: pass k alternates two probe variants (another access order).
usage: mallprobe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi
from finmlkit_amd._ffi import DeviceArray
ctx = _ffi.default_context()
big = DeviceArray(ctx, (8 << 30) // 8, np.float64)
ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(1), C.c_int64((8 << 30) // 4), big.p)
ctx.sync()
ms = C.c_double()
for mb in (16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 8192):
    nbytes = mb << 20
    for variant in (1, 2):
        rates = []
        for k in range(6):
            ctx.call("fmk_diag_read_bandwidth", big.p, C.c_size_t(nbytes), C.c_int(variant), C.c_int(8), C.byref(ms))
            rates.append(nbytes / ms.value / 1e9)      # bytes per ms / 1e9 = TB/s
        print("%5d MB  variant %d   TB/s per pass: %s" % (mb, variant, " ".join("%5.2f" % r for r in rates)), flush=True)
