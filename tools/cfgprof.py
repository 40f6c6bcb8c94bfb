#!/usr/bin/env python3
"""ONE secondary config of bench.py, alone in a process, for the offline profiles behind `other_configs.roofline`:
    tools/cfgprof.py KEY [ticks] [reps]
runs the same library call(s) bench.py times under KEY, `reps` times after one untimed call, between two marker kernels
(k_diag_marker) so that tools/cfgprof_summarize.py can cut the set-up (synthesis, thresholds) out of a rocprofv3 trace.
Under `rocprofv3 --kernel-trace --stats` -> profiles/r06_KEY_kernel_stats.csv; under `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
(separate runs) -> the traffic of profiles/traffic_other_configs.json."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64

KEYS = ("cfg3_volume_index", "cfg3_volume_build_ohlcv", "cfg3_dollar_index", "cfg3_dollar_build_ohlcv", "cfg4_equal_bars", "cfg4_lognormal",
        "cfg4_equal_bars_full_mantissa", "cfg4_lognormal_full_mantissa", "lagged_returns_5s", "ewmst_60s",
        "cusum_floor_5e-4", "cusum_floor_1e-5")


def main():
    key = sys.argv[1]
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**9
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    assert key in KEYS, KEYS
    ctx = _ffi.default_context()
    t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
    clock, ci = t.time_bar_index(60.0)
    if key.startswith("cfg3"):
        o = t.bar_ohlcv(ci, want_median=False)
        vol_total = float(o["volume"].to_host().astype(np.float64).sum())
        span_days = (t.first_last_ts()[1] - t.first_last_ts()[0]) / 86400e9
        vthr = vol_total / max(span_days, 1e-9) / 2000.0
        dthr = vthr * float(np.median(o["close"].to_host()))
        del o
        fn = {"cfg3_volume_index": lambda: t.volume_bar_index(vthr),
              "cfg3_volume_build_ohlcv": lambda: t.bar_ohlcv(t.volume_bar_index(vthr), want_median=True),
              "cfg3_dollar_index": lambda: t.dollar_bar_index(dthr),
              "cfg3_dollar_build_ohlcv": lambda: t.bar_ohlcv(t.dollar_bar_index(dthr), want_median=True)}[key]
    elif key.startswith("cfg4"):
        tt, cc = t, ci
        if "full_mantissa" in key:
            am2 = DeviceArray(ctx, n, np.float32)
            ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(42), c_i64(n), am2.p)
            tt = engine.DeviceTrades(ctx, t.ts, t.price, am2, t.side)
        if "lognormal" in key:
            rng = np.random.default_rng(7)
            lens = np.maximum(1, rng.lognormal(np.log(1200.0) - 0.5, 1.0, int(n / 1200 * 1.3)).astype(np.int64))
            ci_h = np.concatenate([[-1], np.cumsum(lens) - 1])
            cc = DeviceArray.from_host(ctx, ci_h[ci_h <= n - 1].astype(np.int64))
        fn = lambda: tt.bars_fused(cc, 0.01, 3.0)                    # noqa: E731
    elif key.startswith("cusum"):
        # bench.py's CUSUM rows: sigma = ewmst(60 s) of the 5 s lagged returns, sigma_mult 2; the floor decides the tier (5e-4: the
        # chain walk of fmk_cusum_chain.hip; 1e-5: the one-pass form of fmk_cusum_onepass.h)
        floor = float(key.split("_")[-1])
        ret = t.lagged_returns(5.0, True)
        sig = t.ewmst(ret, 60.0)
        del ret
        cus = DeviceArray(ctx, 8_000_000 if n >= 1_000_000_000 else max(n, 16), np.int64)
        m, rounds = c_i64(), c_i64()
        fn = lambda: ctx.call("fmk_cusum_bar_indexer_dev", t.ts.p, t.price.p, sig.p, c_i64(n), C.c_double(floor), C.c_double(2.0),   # noqa: E731
                              cus.p, c_i64(cus.n), C.byref(m), C.byref(rounds))
    else:
        ret = t.lagged_returns(5.0, True)
        fn = (lambda: t.lagged_returns(5.0, True)) if key == "lagged_returns_5s" else (lambda: t.ewmst(ret, 60.0))
    r = fn()
    del r
    ctx.sync()
    ctx.call("fmk_diag_marker_dev", C.c_int(1))
    ms = []
    for _ in range(reps):
        ctx.timer_start()
        r = fn()
        ms.append(ctx.timer_stop())
        del r
    ctx.call("fmk_diag_marker_dev", C.c_int(2))
    ctx.sync()
    print(f"CFGPROF {key} n={n} reps={reps} device_ms={min(ms):.4f} (all: {' '.join('%.3f' % m for m in ms)})", flush=True)


if __name__ == "__main__":
    main()
