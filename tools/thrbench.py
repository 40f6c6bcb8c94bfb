#!/usr/bin/env python3
"""Volume / dollar bar indexer time against mean bar length at N ticks (which tier of fmk_volume.hip / fmk_dollar.hip
serves which length).   usage: thrbench.py [N] [L1,L2,...] [volume,dollar]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
ctx.set_fast_threshold(True)       # time the parallel algorithms; uncertified decisions are printed, not redone
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
probe = engine.DeviceTrades.synth(1_000_000, seed=42, ctx=ctx)
mean_v = float(probe.amount.to_host().astype(np.float64).mean())
mean_d = float((probe.amount.to_host().astype(np.float64) * probe.price.to_host()).mean())
print(f"n={n} mean amount {mean_v:.4f} mean dollars {mean_d:.2f}")
LENGTHS = [int(float(x)) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else \
    [300, 865, 1500, 2500, 3500, 5000, 8000, 12000, 30000, 100000, 1000000]
KINDS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["volume", "dollar"]
for L in LENGTHS:
    row = []
    for kind, thr in (("volume", mean_v * L), ("dollar", mean_d * L)):
        if kind not in KINDS:
            continue
        fn = t.volume_bar_index if kind == "volume" else t.dollar_bar_index
        ci = fn(thr); ctx.sync()
        ms = []
        for _ in range(3):
            ctx.timer_start(); ci = fn(thr); ms.append(ctx.timer_stop())
        row.append("%s %8.2f ms (%9d bars, %d uncertified)" % (kind, min(ms), ci.n - 1, t.last_uncertified))
        del ci
    print("mean bar %8d ticks: %s" % (L, "   ".join(row)), flush=True)
