#!/usr/bin/env python3
"""Dollar bars at N ticks in the library's DEFAULT (exact) mode -- closed form + exact tier (csrc/fmk_dollar_exact.hip) --
against the fast mode (closed form alone, decisions inside the reference's rounding drift only counted) and, on a prefix, the
sequential oracle.   usage: dlexact.py [N] [prefix] [mean bar length, ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from oracle import oracle as orc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
prefix = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
lengths = [float(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [864.6]
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
probe = engine.DeviceTrades.synth(1_000_000, seed=42, ctx=ctx)
mean_d = float((probe.amount.to_host().astype(np.float64) * probe.price.to_host()).mean())
orc.build()
ts, px, am, sd = orc.synth(42, 0, min(prefix, n))
for L in lengths:
    thr = mean_d * L
    def timed(fn, reps=3):
        best = None
        for _ in range(reps):
            ctx.sync(); t0 = time.perf_counter(); r = fn(); ctx.sync()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return best, r
    ms_exact, exact = timed(lambda: t.dollar_bar_index(thr))
    unc_exact = t.last_uncertified
    ctx.set_fast_threshold(True)
    try:
        ms_fast, fast = timed(lambda: t.dollar_bar_index(thr))
        unc_fast = t.last_uncertified
    finally:
        ctx.set_fast_threshold(False)
    e, f = exact.to_host(), fast.to_host()
    ndiff = int((e != f).sum()) if len(e) == len(f) else -1
    t0 = time.perf_counter()
    want = orc._dollar_bar_indexer(px, am, thr)
    t_orc = time.perf_counter() - t0
    k = len(want)
    ok_exact = bool(np.array_equal(e[:k], want)) and (len(e) == k or e[k] >= len(px))
    dfast = int((f[:k] != want).sum())
    print(f"n={n:.3g} mean bar {L:g} ticks thr={thr!r}: {len(e) - 1} bars | exact mode {ms_exact:.2f} ms, uncertified {unc_exact}"
          f" | fast mode {ms_fast:.2f} ms, uncertified {unc_fast} | closes that differ exact vs fast: {ndiff}"
          f" | oracle on the first {len(px):.3g} ticks ({t_orc:.1f} s, {k - 1} closes): exact mode {'EQUAL' if ok_exact else 'DIFFERENT'},"
          f" fast mode differs at {dfast} closes", flush=True)
