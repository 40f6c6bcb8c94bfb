#!/usr/bin/env python3
"""Randomized sweep of the sharded time-bar step (finmlkit_amd/dist.py, what bench.py runs at --gpus N > 1) with virtual
ranks on one device: random world sizes, ticks per rank, stream density and bar interval; the concatenated per-rank outputs
must equal the un-sharded run bit for bit.   usage: fuzz_sharded.py [configs] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_dist import KEYS, _run_sharded


def sweep(configs, seed, verbose=True):
    from finmlkit_amd import _ffi, dist, engine
    ctx = _ffi.default_context()
    rng = np.random.default_rng(seed)
    fails = []
    ran = 0
    for it in range(configs):
        world = int(rng.integers(2, 9))
        n = int(rng.choice([7_001, 20_000, 65_537, 131_072, 250_000, 400_003]))
        sparse = bool(rng.random() < 0.3)
        gap = engine.SPARSE_GAP_MOD if sparse else engine.DENSE_GAP_MOD
        # ticks per second of the stream, from a probe, to pick intervals that leave every shard a few complete bars
        probe = engine.DeviceTrades.synth(n, seed=42, gap_mod=gap, ctx=ctx)
        t0, t1 = probe.first_last_ts()
        span_s = (t1 - t0) / 1e9
        del probe
        cands = [iv for iv in (1.0, 5.0, 60.0, 300.0, 3600.0) if span_s / iv >= 4]
        if not cands:
            continue
        interval = float(rng.choice(cands))
        median = bool(rng.random() < 0.7)
        name = f"world={world} n/rank={n} sparse={sparse} interval={interval} median={median}"
        ran += 1
        try:
            clock, got = _run_sharded(engine, dist, ctx, world, n, gap, interval, want_median=median, steps=int(rng.integers(1, 3)))
            whole = engine.DeviceTrades.synth(world * n, seed=42, gap_mod=gap, ctx=ctx)
            wclock, wci = whole.time_bar_index(interval)
            want = engine.to_host(whole.bar_ohlcv(wci, want_median=median))
            np.testing.assert_array_equal(clock, wclock.to_host(), err_msg="clock")
            for k in KEYS:
                if k in got:
                    np.testing.assert_array_equal(got[k], want[k], err_msg=k)
            del whole
        except Exception as e:      # noqa: BLE001
            fails.append(f"[seed {seed} config {it}] {name}: {type(e).__name__}: {' '.join(str(e).split())[:200]}")
            if verbose:
                print(fails[-1], flush=True)
        ctx.trim()
    return fails, ran


if __name__ == "__main__":
    cfgs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f, ran = sweep(cfgs, seed)
    print(f"{cfgs} configurations drawn, {ran} run, seed {seed}: {len(f)} failures")
    sys.exit(1 if f else 0)
