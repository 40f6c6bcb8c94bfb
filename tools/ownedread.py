#!/usr/bin/env python3
"""Price / amount / side streamed by one wave per segment with lane l owning R consecutive ticks (fmk_diag_read_owned): does the
memory pipeline deliver 13 B/tick as fast when lanes own consecutive ticks as in the chunk layout (R = 1)?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finmlkit_amd import _ffi, engine
ctx = _ffi.default_context()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
t = engine.DeviceTrades.synth(n, seed=42)
ctx.sync()
for seg in (1200, 1201, 1337):
    for r in (1, 2, 4, 8):
        for bpc in (4, 6, 8):
            ms = C.c_double()
            best = 1e9
            for _ in range(4):
                ctx.call("fmk_diag_read_owned", t.price.p, t.amount.p, t.side.p, C.c_int64(n), C.c_int(seg), C.c_int(r), C.c_int(bpc),
                         C.byref(ms))
                best = min(best, ms.value)
            print(f"seg {seg:5d}  R {r}  {bpc} blocks/CU: {best:7.3f} ms  {13.0 * n / best / 1e6:8.1f} GB/s", flush=True)
