#!/usr/bin/env python3
"""Every resident path at the reference's benchmark size (39 M ticks, one-minute bars of ~877 ticks) against the same path at 1e9 ticks:
a path whose time at 39 M is far more than 3.9 % of its time at 1e9 carries a fixed cost (host round trips, launch-bound phases).
Device time by the context's HIP-event timer, best of 5, columns resident in HBM.   usage: sizescale.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64
ctx = _ffi.default_context()


def best(fn, reps=5):
    fn(); ctx.sync()
    ms = []
    for _ in range(reps):
        ctx.timer_start(); r = fn(); ms.append(ctx.timer_stop()); del r
    return min(ms)


def table(n):
    gap_mod = int(2 * 44_640 * 60e9 / 39_171_929)                    # tools/apibench.py's tape: ~877 ticks per minute
    t = engine.DeviceTrades.synth(n, seed=42, gap_mod=gap_mod, ctx=ctx)
    clock, ci = t.time_bar_index(60.0)
    o = t.bar_ohlcv(ci, want_median=False)
    vol_total = float(o["volume"].to_host().astype(np.float64).sum())
    nb = int(ci.n) - 1
    vthr = vol_total / nb                                            # as many volume / dollar bars as minutes
    dthr = vthr * float(np.median(o["close"].to_host()))
    del o
    out = {}
    out["time_bar_index"] = best(lambda: t.time_bar_index(60.0))
    out["time_bars_ohlcv+median"] = best(lambda: t.time_bars_ohlcv(60.0))
    out["tick_bar_index"] = best(lambda: t.tick_bar_index(877))
    out["volume_bar_index"] = best(lambda: t.volume_bar_index(vthr))
    out["dollar_bar_index"] = best(lambda: t.dollar_bar_index(dthr))
    out["bar_directional"] = best(lambda: t.bar_directional(ci))
    out["bars_fused (cfg 4)"] = best(lambda: t.bars_fused(ci, 0.01, 3.0))
    ret = t.lagged_returns(5.0, True)
    out["lagged_returns"] = best(lambda: t.lagged_returns(5.0, True))
    sig = t.ewmst(ret, 60.0)
    out["ewmst"] = best(lambda: t.ewmst(ret, 60.0))
    cus = DeviceArray(ctx, max(n // 100, 1 << 20), np.int64)
    m, rounds = c_i64(), c_i64()
    for floor in (5e-4, 1e-5):
        out[f"cusum floor {floor:g}"] = best(lambda: ctx.call("fmk_cusum_bar_indexer_dev", t.ts.p, t.price.p, sig.p, c_i64(n), C.c_double(floor),
                                                               C.c_double(2.0), cus.p, c_i64(cus.n), C.byref(m), C.byref(rounds)))
    return out, nb


small, nb_s = table(39_171_929)
big, nb_b = table(1_000_000_000)
print(f"bars: {nb_s} at 39 M ticks, {nb_b} at 1e9")
print("%-26s %10s %10s %9s" % ("path", "39 M (ms)", "1e9 (ms)", "ratio / 0.0392"))
for k in small:
    print("%-26s %10.3f %10.3f %9.2f" % (k, small[k], big[k], small[k] / big[k] / 0.0391719))
