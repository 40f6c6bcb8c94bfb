import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import c_i64
ctx = _ffi.default_context()
n = 200_000_000
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
for iv in (60.0, 3600.0, 86400.0):
    clock, ci = t.time_bar_index(iv)
    for _ in range(2):
        ctx.sync(); t0 = time.perf_counter(); r = t.bar_directional(ci); ctx.sync(); dt = (time.perf_counter() - t0) * 1e3
    st = (c_i64 * 10)(); ctx.call("fmk_diag_dir_redo", st)
    print(f"iv {iv}: {dt:.2f} ms, redo {st[0]} bars {st[2]}/{st[1]} slow tiles, per column {list(st[3:10])}  force={os.environ.get('FMK_DIR_FORCE_REDO')} rows={os.environ.get('FMK_DIR_REDO_ROWS')}", flush=True)
