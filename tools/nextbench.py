#!/usr/bin/env python3
"""Full-size timings of the SURVEY 8(f) "next" rows that bench.py does not time: comp_bar_trade_size_features and the rolling volume
profile on the 1-minute bars / footprints of N resident ticks, comp_trade_side_vector, merge_split_trades.  usage: nextbench.py [N]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from finmlkit_amd._ffi import DeviceArray, c_i64, c_f64

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = _ffi.default_context()
t = engine.DeviceTrades.synth(n, seed=42, ctx=ctx)
clock, ci = t.time_bar_index(60.0)
nb = ci.n - 1
o = t.bar_ohlcv(ci, want_median=True)


def best(fn, reps=3):
    fn(); ctx.sync()
    b = 1e9
    for _ in range(reps):
        ctx.timer_start(); fn(); b = min(b, ctx.timer_stop())
    return b


# ---- comp_bar_trade_size_features: theta = each bar's median trade size (what BarBuilderBase.build_trade_size_features passes)
theta = o["median_trade_size"]
keys = ("mean_size_rel", "size_95_rel", "pct_block", "size_gini")
ts_out = {k: DeviceArray(ctx, nb, np.float32) for k in keys}
ms = best(lambda: ctx.call("fmk_comp_bar_trade_size_dev", t.amount.p, C.c_int(t.amount_is_f64), c_i64(t.n), theta.p, ci.p,
                           c_i64(ci.n), c_f64(5.0), *[ts_out[k].p for k in keys]))
print(f"comp_bar_trade_size_features  {nb} bars of {n // nb} ticks: {ms:8.2f} ms  ({4 * n / ms / 1e6:7.0f} GB/s of the 4 B/tick read)", flush=True)

# ---- rolling volume profile over the bars' footprints (30-minute window, 0.01 price tick, no bucketing / 50 bins)
off, flat, bar, bad = t.bar_footprints(ci, o["low"], o["high"], 0.01)
bts = t.gather_ts(ci)
n_lev = int(flat["price_levels"].n)
first_bar = int(np.searchsorted(bts.to_host()[1:], bts.to_host()[1] + 30 * 60 * 10**9))
outs = [DeviceArray(ctx, nb, np.int32) for _ in range(3)] + [DeviceArray(ctx, nb, np.float32)]
ts_bars = DeviceArray.from_host(ctx, np.ascontiguousarray(bts.to_host()[1:]))
for bins in (-1, 50):
    ms = best(lambda: ctx.call("fmk_volume_profile_rolling_dev", ts_bars.p, o["high"].p, o["low"].p, off.p, flat["price_levels"].p,
                               flat["buy_volumes"].p, flat["sell_volumes"].p, c_i64(nb), c_i64(first_bar), c_i64(30 * 60 * 10**9),
                               c_i64(bins), c_f64(0.01), c_f64(0.68), *[x.p for x in outs]))
    print(f"volume_profile_rolling        {nb} bars, {n_lev} levels, 30-bar windows, n_bins {bins}: {ms:8.2f} ms", flush=True)

# ---- the two loops of TradesData(preprocess=True)
side = DeviceArray(ctx, n, np.int8)
ms = best(lambda: ctx.call("fmk_comp_trade_side_vector_dev", t.price.p, c_i64(n), side.p))
print(f"comp_trade_side_vector        {n} ticks: {ms:8.2f} ms", flush=True)
ibm = DeviceArray(ctx, n, np.uint8); ibm.zero()
m = c_i64()
o_ts, o_px, o_am, o_sd = (DeviceArray(ctx, n, d) for d in (np.int64, np.float64, np.float32, np.int8))
ms = best(lambda: ctx.call("fmk_merge_split_trades_dev", t.ts.p, t.price.p, t.amount.p, ibm.p, c_i64(n), o_ts.p, o_px.p, o_am.p,
                           o_sd.p, c_i64(n), C.byref(m)))
print(f"merge_split_trades (fill)     {n} ticks -> {m.value}: {ms:8.2f} ms", flush=True)
