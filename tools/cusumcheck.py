#!/usr/bin/env python3
"""Chain walk vs fixed point vs the sequential C oracle for _cusum_bar_indexer on the synthetic stream (sigma from the device's
ewmst, downloaded, so all three see the same input).  usage: cusumcheck.py N floor [floor ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64, c_f64
from oracle import oracle as orc
n = int(float(sys.argv[1]))
floors = [float(x) for x in sys.argv[2:]]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
ret = t.lagged_returns(5.0, True)
sig = t.ewmst(ret, 60.0)
del ret
ts, px, sg = t.ts.to_host(), t.price.to_host(), sig.to_host()
out = DeviceArray(ctx, n, np.int64)
m, rounds = c_i64(), c_i64()
tier, opened, status = c_i64(), c_i64(), c_i64()


def dev(mode):
    os.environ["FMK_CUSUM_CHAIN"] = mode
    ctx.call("fmk_cusum_bar_indexer_dev", t.ts.p, t.price.p, sig.p, c_i64(n), c_f64(floor), c_f64(2.0), out.p, c_i64(n),
             C.byref(m), C.byref(rounds))
    _ffi.lib().fmk_diag_cusum_last(C.byref(tier), C.byref(opened), C.byref(status))
    seg, rate = c_i64(), c_f64()
    _ffi.lib().fmk_diag_cusum_segments(C.byref(seg), C.byref(rate))
    return out.view(0, m.value).to_host().copy(), (tier.value, opened.value, status.value, seg.value)


for floor in floors:
    want = orc._cusum_bar_indexer(ts, px, sg.copy(), floor, 2.0)
    for mode in ("0", "2"):
        for joint in ("0", "1") if mode == "2" else ("0",):
            got, info = dev(mode if joint == "0" else mode + ":joint=1")
            k = min(len(got), len(want))
            bad = np.flatnonzero(got[:k] != want[:k])
            print(f"floor {floor}: mode {mode} joint {joint} tier/opened/status {info}: {len(got)} vs oracle {len(want)} closes,",
                  "identical" if len(got) == len(want) and bad.size == 0 else f"FIRST DIFFERENCE at entry {bad[0] if bad.size else k}: "
                  f"got {got[bad[0]:bad[0] + 3] if bad.size else got[k:k + 3]} want {want[bad[0]:bad[0] + 3] if bad.size else want[k:k + 3]}", flush=True)
