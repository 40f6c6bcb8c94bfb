#!/usr/bin/env python3
"""Where the waves of the one-pass cfg-4 kernel spend their cycles (a -DFU_TIMING build: tools/ab_variant.sh timing "-DFU_TIMING"):
    python tools/ab_lib.py finmlkit_amd/lib/ab/libfmk_hip_timing.so tools/fu_phases.py [N]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
clock, ci = t.time_bar_index(60.0)
am2 = DeviceArray(ctx, n, np.float32)
ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(42), c_i64(n), am2.p)
t2 = engine.DeviceTrades(ctx, t.ts, t.price, am2, t.side)
out = (c_i64 * 4)()
for name, tr in (("dyadic (histogram variant)", t), ("full-mantissa (float64-volume variant)", t2)):
    r = tr.bars_fused(ci, 0.01, 3.0); del r
    ctx.call("fmk_diag_fused_phases", out)
    r = tr.bars_fused(ci, 0.01, 3.0); del r
    ctx.call("fmk_diag_fused_phases", out)
    nb = ci.n - 1
    med = (out[1] - out[3]) % (1 << 64)
    print(f"{name}: per bar and wave, shader cycles: loads + walk {out[0] / nb:9.0f}   fold + outputs {(out[2] - med) / nb:9.0f}   median {med / nb:9.0f}")
