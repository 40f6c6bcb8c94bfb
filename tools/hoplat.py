#!/usr/bin/env python3
"""Cost of a dependent far access for ONE wave (what the volume chain walk pays per close): hops over an 8 GB array of
doubles, next address depending on the data, for several strides and numbers of coalesced 512 B rows per hop."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi
from finmlkit_amd._ffi import DeviceArray
ctx = _ffi.default_context()
n = 1_000_000_000
buf = DeviceArray(ctx, n, np.float64)
buf.zero(); ctx.sync()
print("stride (elements / bytes)   rows per hop   cycles per hop   ns per hop")
for stride in (64, 512, 5000, 40000, 1_000_003):
    for loads in (1, 8):
        hops = 20000
        cyc, ms = C.c_double(), C.c_double()
        for _ in range(2):
            ctx.call("fmk_diag_hop_latency", buf.p, C.c_int64(n), C.c_int64(stride), C.c_int(loads), C.c_int(hops),
                     C.byref(cyc), C.byref(ms))
        print("%10d / %-10d %8d %16.0f %12.0f" % (stride, stride * 8, loads, cyc.value, ms.value * 1e6 / hops), flush=True)
