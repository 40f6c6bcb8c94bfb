#!/usr/bin/env python3
"""Known byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: the
factor 2 on FETCH_SIZE is documented for 16 B/lane reads only).  Launches, ONCE each, over a 4 GB buffer (far beyond the
256 MiB Infinity Cache): 16 B/lane reads, 8 B/lane reads (what the bar reducers issue per price), 4 B/lane reads (per
float32 amount), 8 B/lane stores -- and then ONE launch of the dominant kernel of bench.py (k_bar_ohlcv_small, fused median)
at N ticks.  Run it under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`
(tools/pmc_calibrate.sh); tools/pmc_summarize.py turns the two counter files into profiles/traffic_constants.json.
usage: pmc_calibrate.py [N ticks] [probe bytes]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
nbytes = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000_000
ctx = _ffi.default_context()
buf = DeviceArray(ctx, nbytes // 8, np.int64)
buf.zero(); ctx.sync()
ms = C.c_double()
for variant in (0, 1, 2, 3):
    ctx.call("fmk_diag_read_bandwidth", buf.p, C.c_size_t(nbytes), C.c_int(variant), C.c_int(8), C.byref(ms))
    print(f"probe variant {variant}: {nbytes} bytes, {ms.value:.3f} ms", flush=True)
del buf
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
clock, ci = t.time_bar_index(60.0)
out = t.alloc_ohlcv(ci.n - 1, True)
t.bar_ohlcv(ci, want_median=True, out=out)
ctx.sync()
print(f"dominant kernel: n={n} bars={ci.n - 1}", flush=True)
