#!/usr/bin/env python3
"""merge_split_trades (utils.py:263-329) on N resident ticks whose timestamps repeat (the synthetic stream's print blocks): ms for the
count call and for the fill call.  usage: mergebench.py [N]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
ibm = DeviceArray(ctx, n, np.uint8)
ibm.zero()
m = c_i64()
o_ts, o_px, o_am, o_sd = DeviceArray(ctx, n, np.int64), DeviceArray(ctx, n, np.float64), DeviceArray(ctx, n, np.float32), DeviceArray(ctx, n, np.int8)
for rep in range(3):
    ctx.timer_start()
    ctx.call("fmk_merge_split_trades_dev", t.ts.p, t.price.p, t.amount.p, ibm.p, c_i64(n), None, None, None, None, c_i64(0), C.byref(m))
    a = ctx.timer_stop()
    ctx.timer_start()
    ctx.call("fmk_merge_split_trades_dev", t.ts.p, t.price.p, t.amount.p, ibm.p, c_i64(n), o_ts.p, o_px.p, o_am.p, o_sd.p, c_i64(n), C.byref(m))
    b = ctx.timer_stop()
    print("merge_split_trades: count %.2f ms, fill %.2f ms; %d -> %d trades (checksum %d)" %
          (a, b, n, m.value, int(o_ts.view(0, min(m.value, 1_000_000)).to_host().sum() % 1000003)), flush=True)
