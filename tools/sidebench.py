#!/usr/bin/env python3
"""comp_trade_side_vector (the tick rule with forward fill, utils.py) on N resident ticks: ms per call.  usage: sidebench.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
out = DeviceArray(ctx, n, np.int8)
ms = []
for _ in range(4):
    ctx.timer_start()
    ctx.call("fmk_comp_trade_side_vector_dev", t.price.p, c_i64(n), out.p)
    ms.append(ctx.timer_stop())
h = out.view(0, 1_000_000).to_host()
print("comp_trade_side_vector: %.2f ms per %d ticks (checksum of the first 1e6 sides %d)" % (min(ms[1:]), n, int(h.astype(np.int64).sum())))
