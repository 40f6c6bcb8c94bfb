#!/usr/bin/env python3
"""Which allocation does the step time depend on?  One process, four phases of R rounds each; a round re-creates
(inputs?, per-bar buffers?) and times 20 steps of time-bar OHLCV + median:
    A keep both      B new per-bar buffers (close indices, outputs)      C new input columns      D both new
`new` = drop the arrays, give the pooled blocks back to the driver (ctx.trim), allocate and fill again.
usage: allocwarm.py [N] [rounds]"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _ffi.default_context()
S = {}


def inputs():
    S.pop("t", None); gc.collect(); ctx.trim()
    S["t"] = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)


def perbar():
    for k in ("ci", "o", "clock"):
        S.pop(k, None)
    gc.collect(); ctx.trim()
    S["clock"], S["ci"] = S["t"].time_bar_index(60.0)
    S["o"] = S["t"].alloc_ohlcv(S["ci"].n - 1, True)


def measure():
    t, ci, o = S["t"], S["ci"], S["o"]
    for _ in range(3):
        t.bar_ohlcv(ci, True, out=o)
    ctx.sync()
    ms = []
    for _ in range(20):
        ctx.timer_start(); t.bar_ohlcv(ci, True, out=o); ms.append(ctx.timer_stop())
    return float(np.mean(ms))


inputs(); perbar()
for phase, (ni, nb) in (("A keep both", (0, 0)), ("B new per-bar buffers", (0, 1)), ("C new inputs", (1, 0)),
                        ("D both new", (1, 1)), ("A keep both (again)", (0, 0))):
    out = []
    for r in range(rounds):
        if ni:
            inputs()
            if not nb:                     # the per-bar buffers stay; nothing else to do
                pass
        if nb:
            perbar()
        out.append(measure())
    print("%-24s %s   spread %.1f %%" % (phase, " ".join("%.3f" % x for x in out), 100 * (max(out) - min(out)) / min(out)),
          flush=True)
