#!/usr/bin/env python3
"""_cusum_bar_indexer at N ticks on the synthetic stream with an EWM sigma of 5 s log returns (the QuickStart flow):
total time, rounds of the fixed point (or chunks opened by the chain walk), closes and the tier that answered, for a range
of sigma floors from 1e-5 (thresholds reached every 6-210 ticks) to the reference kit's default 5e-4 (on this quiet tape
the floor IS the threshold: one close per 2.4e5 ticks).  FMK_CUSUM_CHAIN=0 times the fixed point alone.
usage: cusumbench.py [N] [floor ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64, c_f64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
floors = [float(x) for x in sys.argv[2:]] or [1e-5, 5e-4]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
ret = t.lagged_returns(5.0, True)
sig = t.ewmst(ret, 60.0)
del ret
m, rounds = c_i64(), c_i64()
tier, opened, status = c_i64(), c_i64(), c_i64()
out = DeviceArray(ctx, n, np.int64)
for floor in floors:
    for rep in range(3):
        ctx.timer_start()
        ctx.call("fmk_cusum_bar_indexer_dev", t.ts.p, t.price.p, sig.p, c_i64(n), c_f64(floor), c_f64(2.0), out.p, c_i64(n),
                 C.byref(m), C.byref(rounds))
        ms = ctx.timer_stop()
        _ffi.lib().fmk_diag_cusum_last(C.byref(tier), C.byref(opened), C.byref(status))
        seg, rate = c_i64(), c_f64()
        _ffi.lib().fmk_diag_cusum_segments(C.byref(seg), C.byref(rate))
        u, fl, pf, ch = c_i64(), c_i64(), c_i64(), c_i64()
        _ffi.lib().fmk_diag_cusum_onepass(C.byref(u), C.byref(fl), C.byref(pf), C.byref(ch))
        print("   one pass: used %d, fix-up launches %d, chunks pending after the first %d of %d" % (u.value, fl.value, pf.value, ch.value))
        print("sigma_floor %g: %.2f ms  rounds %d  closes %d  tier %d (walk: %d opened / events, status %d, %d later segments, rate %.3f)  checksum %d" %
              (floor, ms, rounds.value, m.value, tier.value, opened.value, status.value, seg.value, rate.value,
               int(out.view(0, m.value).to_host().sum())), flush=True)
