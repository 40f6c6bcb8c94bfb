#!/usr/bin/env python3
"""The reference's one published benchmark through THIS build's API (examples/PerformanceTest.ipynb cells 12-14: 39 171 929 trades
-> 44 640 one-minute bars via `TimeBarKit(trades, pd.Timedelta(minutes=1)).build_ohlcv()`, 1.896 s cold / 0.1728 s warm with Numba
on the author's machine).  Same call, host-resident NumPy columns (float32 amounts, as TradesData(preprocess=True) leaves them), so
the time INCLUDES the host-to-device copy of the columns -- the number the device-resident `bench.py` value leaves out.

    python tools/apibench.py [n_ticks]        -> one JSON line (also used by bench.py's other_configs)
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd

N_REF, BARS_REF = 39_171_929, 44_640


def run(n=N_REF, reps=5, ctx=None):
    from finmlkit_amd import _ffi, engine
    from finmlkit_amd.bar.data_model import TradesData
    from finmlkit_amd.bar.kit import TimeBarKit
    ctx = ctx or _ffi.default_context()
    # 31 days of trades: gaps uniform on 1 .. 2 * 31 d / n ns; generated on the device, then HOST arrays like a loaded file
    gap_mod = int(2 * BARS_REF * 60e9 / N_REF)
    t = engine.DeviceTrades.synth(n, seed=42, gap_mod=gap_mod, ctx=ctx)
    ts, px, am, sd = t.to_numpy()
    del t
    ctx.trim()
    trades = TradesData(ts, px, am, np.arange(n, dtype=np.int64), side=sd, timestamp_unit="ns", preprocess=False)
    assert trades.data["amount"].dtype == np.float32
    times = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        kit = TimeBarKit(trades, period=pd.Timedelta(minutes=1))
        bars = kit.build_ohlcv()
        times.append(time.perf_counter() - t0)
        nb = len(bars)
        del kit
    # the pieces, once more
    t0 = time.perf_counter()
    kit = TimeBarKit(trades, period=pd.Timedelta(minutes=1))
    dev = kit._device()
    ctx.sync()
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    kit._set_bar_close()
    ctx.sync()
    t_idx = time.perf_counter() - t0
    t0 = time.perf_counter()
    kit.build_ohlcv()
    t_build = time.perf_counter() - t0
    up_bytes = sum(c.nbytes for c in (dev.ts, dev.price, dev.amount) + ((dev._side,) if dev._side is not None else ()))
    rate = {}
    try:
        import ctypes as C
        for mode, name in ((1, "hipMemcpy_pinned"), (0, "hipMemcpy_pageable"), (2, "fmk_h2d_columns_pageable")):
            r = C.c_double()
            ctx.call("fmk_diag_h2d_rate", C.c_size_t(256 << 20), C.c_int(mode), C.byref(r))
            rate[name] = round(r.value, 2)
    except AttributeError:
        pass
    return {"n_ticks": n, "n_bars": nb, "cold_ms": times[0] * 1e3, "warm_ms": min(times[1:]) * 1e3,
            "warm_all_ms": [round(x * 1e3, 2) for x in times[1:]],
            "upload_ms": t_up * 1e3, "upload_bytes": up_bytes, "h2d_GBps": up_bytes / t_up / 1e9,
            "index_ms": t_idx * 1e3, "reduce_readback_frame_ms": t_build * 1e3,
            "box_h2d_GBps": rate,
            "reference_published": {"cold_s": 1.8961, "warm_s": 0.1728, "where": "examples/PerformanceTest.ipynb cells 12-14 "
                                    "(Numba, the author's machine, real BTCUSDT trades)"}}


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else N_REF
    print(json.dumps(run(n)))
