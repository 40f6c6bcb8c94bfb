#!/usr/bin/env python3
"""Per-kernel timings of the non-headline configs (BASELINE.json configs[2..3] + tick-level rows).

Not the driver's bench (that is /bench.py, cfg 2).  Prints one JSON line per measured stage:
HIP-event time on the context stream, device-resident inputs, median of `--reps` runs.

    python tools/cfgbench.py --ticks 1000000000
"""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from finmlkit_amd import _ffi, engine  # noqa: E402
from finmlkit_amd._ffi import DeviceArray  # noqa: E402


def timed(ctx, fn, reps):
    fn()
    ctx.sync()
    ts = []
    for _ in range(reps):
        ctx.timer_start()
        r = fn()
        ts.append(ctx.timer_stop())
    return float(np.median(ts)), r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--skip-threshold", action="store_true")
    args = ap.parse_args()
    ctx = _ffi.default_context()
    n = args.ticks
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
    ctx.sync()

    def emit(stage, ms, bytes_per_tick, **kw):
        print(json.dumps({"stage": stage, "ticks": n, "ms": ms, "ticks_per_s": n / (ms * 1e-3),
                          "alg_GBps": bytes_per_tick * n / (ms * 1e-3) / 1e9 if bytes_per_tick else None, **kw}),
              flush=True)

    clock, ci = t.time_bar_index(60.0)
    nb = ci.n - 1
    ms, _ = timed(ctx, lambda: t.time_bar_index(60.0, out=(clock, ci)), args.reps)
    emit("time_bar_indexer(60s)", ms, None, n_bars=nb)
    out = t.alloc_ohlcv(nb, True)
    ms, _ = timed(ctx, lambda: t.bar_ohlcv(ci, True, out=out), args.reps)
    emit("comp_bar_ohlcv+median (fused)", ms, 12, n_bars=nb)
    ms, _ = timed(ctx, lambda: t.bar_ohlcv(ci, False, out=out), args.reps)
    emit("comp_bar_ohlcv (no median)", ms, 12, n_bars=nb)
    ms, _ = timed(ctx, lambda: t.bar_directional(ci), args.reps)
    emit("comp_bar_directional_features", ms, 13, n_bars=nb)
    ms, r = timed(ctx, lambda: t.bar_footprints(ci, out["low"], out["high"], 0.01, 3.0), args.reps)
    emit("comp_bar_footprints (size+fill, CSR)", ms, 13, n_bars=nb, total_levels=int(r[0].to_host()[-1]))
    del r
    # cfg 3: thresholds derived from the data (median daily volume / 2000, QuickStart cells 52-54)
    if not args.skip_threshold:
        dclock, dci = t.time_bar_index(86400.0)
        dv = engine.to_host(t.bar_ohlcv(dci, False))
        vthr = float(np.median(dv["volume"][:-1])) / 2000.0
        dthr = vthr * float(np.median(dv["close"]))
        ctx.set_fast_threshold(True)     # time the parallel algorithms; the uncertified counts are reported below
        ms, vci = timed(ctx, lambda: t.volume_bar_index(vthr), 1)
        emit("volume_bar_indexer (parallel jump tables)", ms, 4, threshold=vthr, n_bars=vci.n - 1,
             uncertified=t.last_uncertified)
        ms, dci2 = timed(ctx, lambda: t.dollar_bar_index(dthr), 1)
        emit("dollar_bar_indexer (parallel closed form)", ms, 12, threshold=dthr, n_bars=dci2.n - 1,
             uncertified=t.last_uncertified)
        vout = t.alloc_ohlcv(vci.n - 1, True)
        ms, _ = timed(ctx, lambda: t.bar_ohlcv(vci, True, out=vout), args.reps)
        emit("comp_bar_ohlcv+median on volume bars", ms, 12, n_bars=vci.n - 1)
        del vout, vci, dci2
    # tick-level volatility
    ms, ret = timed(ctx, lambda: t.lagged_returns(5.0, True), args.reps)
    emit("comp_lagged_returns(5s, log)", ms, 24)
    ms, _ = timed(ctx, lambda: t.ewmst(ret, 60.0), args.reps)
    emit("ewmst(60s)", ms, 40)


if __name__ == "__main__":
    main()
