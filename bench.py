#!/usr/bin/env python3
"""bench.py -- ticks/sec aggregated to 1-minute time bars on N x MI355X.

Workload (BASELINE.json configs[1], per GPU): 1e9 synthetic ticks (generated on the
device, resident in HBM) -> `TimeBarKit(period=60s).build_ohlcv()` semantics: bar clock + close
indices (_time_bar_indexer), OHLC/volume/VWAP/trade count (comp_bar_ohlcv) and the median trade
size.  One "step" = one full pass of that path over the resident columns; outputs stay on the device.

N > 1, one rank per GPU, no launcher needed: `python3 bench.py --gpus N` makes this process rank 0 and starts ranks
1..N-1 as children of itself (RANK / LOCAL_RANK / WORLD_SIZE and the rendezvous path handed down in the environment).
Started by a launcher that already set WORLD_SIZE (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in
the environment), every process is simply its rank.  Rank r holds ticks [r*n, (r+1)*n) of ONE global stream (weak
scaling); the bar that straddles a shard boundary is stitched by a single neighbour send/recv of the trailing partial
bar's raw ticks per step: ncclSend/ncclRecv of librccl behind the C ABI (fmk_comm_*, csrc/fmk_comm.hip), on its own
stream, ordered against the compute stream by events -- no host synchronisation inside a step (finmlkit_amd/dist.py).

Prints ONE JSON line on rank 0 (see the contract in the task statement); `roofline` describes the
dominant kernel (k_bar_ohlcv: 12 algorithmic B/tick), `cpu_baseline` the scalar C oracle on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HBM bytes per tick / per bar of the dominant kernel: counters cannot be collected inside a normal run, so roofline.traffic is an
# OFFLINE rocprofv3 --pmc measurement of the same kernel, scaled to the run's ticks and bars (roofline.traffic_source says so).
# The constants live in the tracked file profiles/traffic_constants.json (written by tools/pmc_summarize.py from the two counter
# passes of tools/pmc_calibrate.sh, with the commit it was measured at and the FETCH_SIZE / WRITE_SIZE corrections calibrated in
# the SAME passes on known byte counts at 16, 8 and 4 bytes per lane: profiles/r02_pmc_calibration.txt); null for other workloads.
def _traffic_constants():
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_constants.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def kernel_source_sha256():
    """Identity of the dominant kernel's code: SHA-256 over the sources k_bar_ohlcv_small is compiled from.  The traffic constants
    carry the value they were measured at (tools/pmc_summarize.py); a different value here means `traffic` describes older code."""
    import hashlib
    h = hashlib.sha256()
    for f in ("fmk_ohlcv.hip", "fmk_median.h", "fmk_dpp.h", "fmk_common.h", "fmk_pairwise.h", "fmk_f32tie.h", "Makefile"):
        with open(os.path.join(ROOT, "finmlkit_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def csrc_sha256():
    """Identity of ALL kernel sources (the per-config traffic figures of profiles/traffic_other_configs.json carry the value they were
    measured at; tools/cfgprof_summarize.py)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "finmlkit_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def _other_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_other_configs.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def config_roofline(key, kernel_ms, alg_bytes, note=None, extra=None):
    """The roofline object of one secondary config: `kernel_ms` is DEVICE time of the library call(s), a HIP-event pair on the
    context's stream (fmk_timer_start / fmk_timer_stop: from the first enqueue to the last completion, the host round trips
    inside a call included -- what a caller waits for, minus the Python around it); `traffic` and `dominant_kernel` come from the
    OFFLINE rocprofv3 passes of the same call (tools/cfgprof.py: --kernel-trace --stats, and --pmc FETCH_SIZE / WRITE_SIZE in
    their own runs), with the hash of the kernel sources they were measured at."""
    tr = _other_traffic().get(key) or {}
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    stale = (tr.get("csrc_sha256") != csrc_sha256()) if tr else None
    r = {"bound": "hbm", "algorithmic_bytes": alg_bytes, "kernel_ms": kernel_ms, "achieved": achieved, "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         # (summed over the launches of that kernel in one call: where launches overlap on two streams the sum can exceed kernel_ms)
         "dominant_kernel": tr.get("dominant_kernel"), "dominant_kernel_ms_summed": tr.get("dominant_kernel_ms_summed", tr.get("dominant_kernel_ms")),
         "kernels_ms_offline": tr.get("kernels_ms"),
         "traffic": tr.get("traffic_bytes"), "traffic_stale": stale,
         "traffic_source": (f"offline rocprofv3 --pmc FETCH_SIZE (x2, profiles/pmc_calibration.txt) + WRITE_SIZE over the kernels of one call "
                            f"(tools/cfgprof.py {key}; sources sha256 {str(tr.get('csrc_sha256'))[:12]}); kernel stats of the same command: "
                            f"profiles/r06_{key}_kernel_stats.csv") if tr else None}
    if note:
        r["note"] = note
    if extra:
        r.update(extra)
    return r


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ticks", type=int, default=1_000_000_000, help="ticks per GPU")
    ap.add_argument("--interval", type=float, default=60.0)
    ap.add_argument("--no-median", action="store_true", help="skip the median trade size (not the reference path)")
    ap.add_argument("--cpu-sample", type=int, default=200_000_000, help="ticks of the CPU-baseline sample (0: skip)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-extras", action="store_true", help="skip the informational cfg 3 / cfg 4 timings")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the sharded step with 1 rank: RCCL communicator of size 1, the rank is its own neighbour "
                         "(ncclSend/ncclRecv to self), plus the boundary-bar launches -- the per-step overhead of the "
                         "multi-GPU path measured on one GPU")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"], help="halo transport (host: staged, tests)")
    ap.add_argument("--placements", type=int, default=1,
                    help="1 (default): the headline is measured on the input columns AS THE LIBRARY ALLOCATED THEM.  K > 1: the columns go through "
                         "DeviceTrades.place() (finmlkit_amd/engine.py: K positions, one every 16 GiB of ONE allocation, probed with the step) "
                         "BEFORE the timed region and the headline is measured on the chosen position -- a diagnostic, roofline.placement says so")
    ap.add_argument("--placed-probe", type=int, default=7,
                    help="after the timed region (1 GPU): the same step on columns placed by the library's opt-in DeviceTrades.place() with this many "
                         "positions -> roofline.placed (0: skip); never part of `value` / `frac`")
    ap.add_argument("--separate-index", action="store_true",
                    help="the step as two library calls (time-bar indexer kernels, then OHLCV + median) instead of the one-launch "
                         "fmk_time_bars_ohlcv_dev -- for A/B timing")
    return ap.parse_args()


def cpu_baseline(args):
    """C oracle (oracle/fmk_oracle.c) on a bounded sample of the same workload: all host cores, and 1 thread."""
    if args.cpu_sample <= 0:
        return None
    from oracle import oracle as orc
    orc.build()
    m = min(args.cpu_sample, args.ticks)
    ts, px, am, sd = orc.synth(args.seed, 0, m)
    def run(threads):
        os.environ["ORC_THREADS"] = str(threads)     # OpenMP over bars (like the reference's prange), same arithmetic
        t0 = time.perf_counter()
        clock, ci = orc._time_bar_indexer(ts, args.interval)
        orc.comp_bar_ohlcv(px, am, ci, want_median=not args.no_median)
        return time.perf_counter() - t0

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    dt1 = run(1)
    dtn = min(run(cores), run(cores)) if cores > 1 else dt1
    return {"value": m / dtn, "unit": "ticks/s", "cores": cores, "kind": "port",
            "sample": f"first {m} ticks of the same synthetic stream, time-bar indexer + comp_bar_ohlcv"
                      f"{'' if args.no_median else ' + median'} (oracle/fmk_oracle.c, gcc -O2 -fopenmp, bars split over "
                      f"{cores} threads): {dtn:.2f} s; 1 thread: {dt1:.2f} s = {m / dt1:.3g} ticks/s"}


def _time_resample(ctx, timed, clock1, o1, nb1):
    """fmk_resample_bars_dev on resident 1-second bars (segments = floor(ts / 60 s) computed on the host from the clock)."""
    import ctypes as C
    import numpy as np
    from finmlkit_amd._ffi import DeviceArray, c_i64
    ts = clock1.to_host()[1:nb1 + 1]
    key = ts // 60_000_000_000
    seg = np.concatenate([[0], np.flatnonzero(np.diff(key)) + 1, [nb1]]).astype(np.int64)
    G = len(seg) - 1
    d_seg = DeviceArray.from_host(ctx, seg)
    outs = [DeviceArray(ctx, G, dt) for dt in (np.float64, np.float64, np.float64, np.float64, np.float32, np.int64,
                                               np.float32, np.float32, np.uint8)]
    def run():
        ctx.call("fmk_resample_bars_dev", d_seg.p, c_i64(G), o1["open"].p, o1["high"].p, o1["low"].p, o1["close"].p,
                 o1["volume"].p, C.c_int(0), o1["trades"].p, o1["vwap"].p, C.c_int(1), o1["median_trade_size"].p,
                 *[x.p for x in outs])
        return None
    return timed(run)


def other_configs(trades, ctx, args):
    """Informational, AFTER the timed region and outside `value`: one pass each of BASELINE.json configs[2] (volume +
    dollar bar indices, data-derived thresholds) and configs[3] (time bars + order-flow + footprints) on the same
    resident columns, host wall time incl. the size/fill phases.  Never raises: a failure is reported as a string."""
    import ctypes as C
    import numpy as np
    from finmlkit_amd import engine
    from finmlkit_amd._ffi import DeviceArray, c_i64
    out = {}

    dev_ms = {}

    def timed(fn, reps=3, key=None):
        """best-of-`reps` host wall time (ms); with `key`, also the best DEVICE time of the call -- a HIP-event pair on the context's
        stream (fmk_timer_start / _stop) -- kept in dev_ms[key] for the config's roofline object"""
        best = None
        for _ in range(reps):
            ctx.sync()
            t0 = time.perf_counter()
            if key:
                ctx.timer_start()
            r = fn()
            if key:
                d = ctx.timer_stop()
                dev_ms[key] = d if key not in dev_ms else min(dev_ms[key], d)
            ctx.sync()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
            del r
        return best

    def levels_of(fused):
        return int(fused[4]["price_levels"].n)

    try:
        n = trades.n
        clock, ci = trades.time_bar_index(args.interval)
        o = trades.bar_ohlcv(ci, want_median=False)
        vol_total = float(o["volume"].to_host().astype(np.float64).sum())
        span_days = (trades.first_last_ts()[1] - trades.first_last_ts()[0]) / 86400e9
        vthr = vol_total / max(span_days, 1e-9) / 2000.0                 # QuickStart: daily volume / 2000
        dthr = vthr * float(np.median(o["close"].to_host()))
        del o
        # volume bars: the library's default (exact) mode -- fragile decisions on the chain of closes are replayed with the
        # reference's sequential sum, so the count that comes back is 0 unless a replay disagreed
        out["cfg3_volume_bar_index_ms"] = timed(lambda: trades.volume_bar_index(vthr), key="cfg3_volume_index")
        out["cfg3_volume_uncertified"] = int(trades.last_uncertified)
        out["cfg3_n_volume_bars"] = int(trades.volume_bar_index(vthr).n)
        # SURVEY 8(d)'s unit for cfg 3 -- indexer AND reducer, 12 algorithmic B/tick: close indices, then OHLCV + median over them
        out["cfg3_volume_build_ohlcv_ms"] = timed(lambda: trades.bar_ohlcv(trades.volume_bar_index(vthr), want_median=True), key="cfg3_volume_build_ohlcv")
        # dollar bars, default (exact) mode as well: closed form + exact tier (csrc/fmk_dollar_exact.hip: the reference's
        # float64 running sum reconstructed at every bar start, fragile bars replayed) -- n_uncertified comes back 0.  The
        # closed form alone (fmk_ctx_set_fast_threshold(1)) is timed next to it with the count of decisions it cannot certify
        out["cfg3_dollar_bar_index_ms"] = timed(lambda: trades.dollar_bar_index(dthr), key="cfg3_dollar_index")
        out["cfg3_dollar_uncertified"] = int(trades.last_uncertified)
        out["cfg3_dollar_exact"] = out["cfg3_dollar_uncertified"] == 0
        exact_idx = trades.dollar_bar_index(dthr)
        out["cfg3_n_dollar_bars"] = int(exact_idx.n - 1)
        out["cfg3_dollar_build_ohlcv_ms"] = timed(lambda: trades.bar_ohlcv(trades.dollar_bar_index(dthr), want_median=True), key="cfg3_dollar_build_ohlcv")
        ctx.set_fast_threshold(True)
        try:
            out["cfg3_dollar_closed_form_only_ms"] = timed(lambda: trades.dollar_bar_index(dthr))
            out["cfg3_dollar_closed_form_uncertified"] = int(trades.last_uncertified)
            fast_idx = trades.dollar_bar_index(dthr)
            out["cfg3_dollar_closes_differing_exact_vs_closed_form"] = (
                int((fast_idx.to_host() != exact_idx.to_host()).sum()) if fast_idx.n == exact_idx.n else -1)
            del fast_idx
        finally:
            ctx.set_fast_threshold(False)
        del exact_idx
        out["cfg4_ohlcv_directional_footprints_ms"] = timed(lambda: trades.bars_fused(ci, 0.01, 3.0), key="cfg4_equal_bars")
        lv_eq = levels_of(trades.bars_fused(ci, 0.01, 3.0))
        out["cfg4_bytes_per_tick"] = 30      # 13 (order flow + OHLC, one lane per bar) + 4 (median of the amounts) + 13 (footprints)
        # the same pass on amounts with a full random float32 mantissa: the footprint level sums are then inexact in every
        # order and every bar takes the tick-ordered accumulation (the synthetic stream's dyadic amounts all certify for the
        # order-free integer path) -- real trade sizes are like that, so this is the number to expect on real data
        am2 = DeviceArray(ctx, n, np.float32)
        ctx.call("fmk_diag_fill_amounts_dev", C.c_uint64(args.seed), c_i64(n), am2.p)
        t2 = engine.DeviceTrades(ctx, trades.ts, trades.price, am2, trades.side)
        out["cfg4_full_mantissa_amounts_ms"] = timed(lambda: t2.bars_fused(ci, 0.01, 3.0), key="cfg4_equal_bars_full_mantissa")
        # cfg 4 on bars of HEAVY-TAILED lengths (lognormal, mean 1 200 ticks, sigma 1: what real one-minute bars look like;
        # profiles/r03_real_bar_lengths.txt) -- the synthetic clock's own bars are all ~1 200 ticks long
        rng = np.random.default_rng(7)
        lens = np.maximum(1, rng.lognormal(np.log(1200.0) - 0.5, 1.0, int(n / 1200 * 1.3)).astype(np.int64))
        ci_h = np.concatenate([[-1], np.cumsum(lens) - 1])
        ci_r = DeviceArray.from_host(ctx, ci_h[ci_h <= n - 1].astype(np.int64))
        out["cfg4_lognormal_bar_lengths_ms"] = timed(lambda: trades.bars_fused(ci_r, 0.01, 3.0), reps=2)
        out["cfg4_lognormal_bar_lengths_full_mantissa_ms"] = timed(lambda: t2.bars_fused(ci_r, 0.01, 3.0), reps=2, key="cfg4_lognormal_full_mantissa")
        lv_ln, nb_ln = levels_of(t2.bars_fused(ci_r, 0.01, 3.0)), int(ci_r.n) - 1
        del ci_r
        # cfg 4 at the ends of the bar-length axis (hourly / daily bars: a workgroup per bar; tools/intervalbench.py has every length)
        for label, iv in (("hourly", 3600.0), ("daily", 86400.0)):
            _, ci_l = trades.time_bar_index(iv)
            out[f"cfg4_{label}_bars_ms"] = timed(lambda: trades.bars_fused(ci_l, 0.01, 3.0), reps=2)
            out[f"cfg4_{label}_bars_full_mantissa_ms"] = timed(lambda: t2.bars_fused(ci_l, 0.01, 3.0), reps=2)
            del ci_l
        del t2, am2
        # TimeBarReader._resample: 1-second bars of the same stream (built here, not timed) -> 1-minute bars
        clock1, ci1 = trades.time_bar_index(1.0)
        o1 = trades.bar_ohlcv(ci1)
        nb1 = ci1.n - 1
        out["resample_1s_to_1min_ms"], out["resample_n_rows"] = _time_resample(ctx, timed, clock1, o1, nb1), nb1
        del o1, clock1, ci1
        # "next" rows: the QuickStart chain lagged returns -> ewmst -> CUSUM bars (CUSUMBarKit's default sigma_floor 5e-4: on this
        # quiet tape a close per 2.4e5 ticks, served by the chain walk of fmk_cusum_chain.hip; 1e-5: a close per ~200 ticks, the
        # one-pass form of fmk_cusum_onepass.h), and the order-flow features on the 1-second bars of io.py:484 (one lane per bar)
        ret = trades.lagged_returns(5.0, True)
        out["lagged_returns_5s_ms"] = timed(lambda: trades.lagged_returns(5.0, True), key="lagged_returns_5s")
        sig = trades.ewmst(ret, 60.0)
        out["ewmst_60s_ms"] = timed(lambda: trades.ewmst(ret, 60.0), key="ewmst_60s")
        del ret
        cus = DeviceArray(ctx, 8_000_000 if n >= 1_000_000_000 else max(n, 16), np.int64)
        m, rounds = c_i64(), c_i64()
        for key, floor in (("cusum_default_floor_5e-4", 5e-4), ("cusum_floor_1e-5", 1e-5)):
            def run(floor=floor):
                ctx.call("fmk_cusum_bar_indexer_dev", trades.ts.p, trades.price.p, sig.p, c_i64(n), C.c_double(floor),
                         C.c_double(2.0), cus.p, c_i64(cus.n), C.byref(m), C.byref(rounds))
            out[key + "_ms"] = timed(run, key="cusum_floor_" + key.split("_")[-1])
            out[key + "_closes"] = int(m.value) - 1
        del sig, cus
        clock1, ci1 = trades.time_bar_index(1.0)
        out["directional_1s_bars_ms"] = timed(lambda: trades.bar_directional(ci1))
        del clock1, ci1
        # comp_bar_trade_size_features on the 1-minute bars, theta = each bar's median trade size (what the kits pass)
        o60 = trades.bar_ohlcv(ci)
        keys4 = [DeviceArray(ctx, int(ci.n) - 1, np.float32) for _ in range(4)]
        out["trade_size_features_ms"] = timed(lambda: ctx.call(
            "fmk_comp_bar_trade_size_dev", trades.amount.p, C.c_int(trades.amount_is_f64), c_i64(n), o60["median_trade_size"].p,
            ci.p, c_i64(ci.n), C.c_double(5.0), *[k.p for k in keys4]))
        del o60, keys4
        # ... and on 2-minute / 10-minute / hourly bars (2 400 / 12 000 / 72 000 ticks: one wave with five tree levels, eight waves on two
        # of np.sum's chunks, the sub-tree workgroup -- profiles/r03_trade_size.txt)
        for tag, iv in (("2min", 120.0), ("10min", 600.0), ("hourly", 3600.0)):
            _, civ = trades.time_bar_index(iv)
            ov = trades.bar_ohlcv(civ)
            kv = [DeviceArray(ctx, int(civ.n) - 1, np.float32) for _ in range(4)]
            out[f"trade_size_{tag}_bars_ms"] = timed(lambda: ctx.call(
                "fmk_comp_bar_trade_size_dev", trades.amount.p, C.c_int(trades.amount_is_f64), c_i64(n), ov["median_trade_size"].p,
                civ.p, c_i64(civ.n), C.c_double(5.0), *[k.p for k in kv]))
            del civ, ov, kv
        # the reference's ONE published benchmark through this build's API, host-resident NumPy columns, H2D copy included
        # (examples/PerformanceTest.ipynb cells 12-14: 39 171 929 trades -> 44 640 one-minute bars, 0.1728 s warm with Numba)
        try:
            from tools import apibench
            ab = apibench.run(reps=3, ctx=ctx)
            out["api_39M_build_ohlcv_ms"] = ab["warm_ms"]
            out["api_39M_build_ohlcv_cold_ms"] = ab["cold_ms"]
            out["h2d_GBps"] = ab["h2d_GBps"]
            out["h2d_box_GBps"] = ab["box_h2d_GBps"]
            out["api_39M_note"] = ("TimeBarKit(trades, 1 min).build_ohlcv() from host NumPy columns (float32 amounts), wall time incl. "
                                   f"the upload of {ab['upload_bytes'] / 1e6:.0f} MB (timestamp, price, amount) and the DataFrame; "
                                   "reference published 172.8 ms warm / 1896 ms cold (Numba, its author's machine): context, not a "
                                   "same-host comparison")
        except Exception as e:                                           # noqa: BLE001
            out["api_39M_error"] = f"{type(e).__name__}: {e}"
        # ---- roofline objects of the secondary configs (SURVEY 8(d)'s algorithmic bytes; DEVICE time by HIP events, best of the reps)
        nb60 = int(ci.n) - 1
        nvb, ndb = out["cfg3_n_volume_bars"] - 1, out["cfg3_n_dollar_bars"]
        cfg4_bytes = lambda bars, levels: 21 * n + bars * (8 + 60 + 8 + 88 + 8 + 26) + levels * 22     # noqa: E731  close idx, OHLCV + median, order flow, level offsets + per-bar footprint features; 22 B per footprint level (the output dtypes of base.py:675-697)
        alg = {"cfg3_volume_index": 4 * n + 8 * (nvb + 1), "cfg3_volume_build_ohlcv": 12 * n + (8 + 68) * nvb,
               "cfg3_dollar_index": 12 * n + 8 * (ndb + 1), "cfg3_dollar_build_ohlcv": 12 * n + (8 + 68) * ndb,
               "cfg4_equal_bars": cfg4_bytes(nb60, lv_eq), "cfg4_equal_bars_full_mantissa": cfg4_bytes(nb60, lv_eq),
               "cfg4_lognormal_full_mantissa": cfg4_bytes(nb_ln, lv_ln),
               "lagged_returns_5s": 24 * n, "ewmst_60s": 24 * n,
               "cusum_floor_5e-4": 24 * n + 8 * out["cusum_default_floor_5e-4_closes"],
               "cusum_floor_1e-5": 24 * n + 8 * out["cusum_floor_1e-5_closes"]}
        notes = {"cfg3_volume_build_ohlcv": "SURVEY 8(d) prices indexer + reducer as ONE read, 12 B/tick; a threshold indexer cannot emit a close before its chain is resolved back to the tape's first tick, so the reducer's pass is a second read by construction: algorithmic_bytes_two_pass (4 + 12 B/tick) is the algorithm's own compulsory figure, frac_two_pass the fraction against it (DESIGN.md section 3)",
                 "cfg3_dollar_build_ohlcv": "SURVEY 8(d) prices indexer + reducer as ONE read, 12 B/tick; the closes are known only when the carry chain is resolved, so the reducer's pass is a second read by construction: algorithmic_bytes_two_pass (12 + 12 B/tick), frac_two_pass (DESIGN.md section 3)",
                 "cfg4_equal_bars": "ONE pass over the ticks (csrc/fmk_fused.h: k_fu_bars reads price + amount + side once for OHLCV + median + order flow + the footprint histogram; k_fu_emit turns the staged level rows into the CSR rows)",
                 "cfg4_equal_bars_full_mantissa": "sizes with a full float32 mantissa do not certify as whole units: OHLCV + median + order flow from one read (k_fu_bars without its histogram), the footprints by their own tick-ordered sweep (a second read, 13 B/tick)",
                 "cfg4_lognormal_full_mantissa": "the primary cfg 4 figure: bars of lognormal length (sigma 1), sizes with a full float32 mantissa",
                 "cusum_floor_5e-4": "24 B/tick read (ts, price, sigma) + 8 B per close; the device time spans the call's host round trips (the chain walk's launches wait for counts)",
                 "cusum_floor_1e-5": "24 B/tick read (ts, price, sigma) + 8 B per close; one-pass form (csrc/fmk_cusum_onepass.h); the device time spans the call's host round trips",
                 "ewmst_60s": "16 B/tick read (ts, returns) + 8 written", "lagged_returns_5s": "16 B/tick read (ts, price) + 8 written"}
        two_pass = {"cfg3_volume_build_ohlcv": 4 * n + 8 * (nvb + 1) + 12 * n + (8 + 68) * nvb,
                    "cfg3_dollar_build_ohlcv": 12 * n + 8 * (ndb + 1) + 12 * n + (8 + 68) * ndb}
        extras = {k: {"algorithmic_bytes_two_pass": v, "frac_two_pass": v / (dev_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                  for k, v in two_pass.items() if k in dev_ms}
        out["roofline"] = {k: config_roofline(k, dev_ms[k], alg[k], notes.get(k), extras.get(k)) for k in alg if k in dev_ms}
        out["note"] = (f"{n} ticks, 1 GPU, best of 3, host wall time; not part of `value`; cfg3: volume AND dollar in the "
                       "library's default exact mode (n_uncertified == 0: provably the reference's close indices)")
    except Exception as e:                                               # noqa: BLE001 -- informational only
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def _probe_step_kernel_ms(ctx, fn, steps):
    """Kernel time per step (ms) of `steps` calls of fn: the library's HIP-event pairs around every launch of the dominant kernel,
    summed per step (the pipelined step launches it twice)."""
    import ctypes as C
    ctx.sync()
    ctx.call("fmk_profile_enable", C.c_int(1))
    for _ in range(steps):
        fn()
    k = (C.c_double * 256)()
    kn = C.c_int()
    ctx.call("fmk_profile_read", k, C.c_int(256), C.byref(kn))
    ctx.call("fmk_profile_enable", C.c_int(0))
    return sum(k[i] for i in range(kn.value)) / steps


def choose_placement(ctx, trades, args, rank, n, step_of, positions):
    """The library's opt-in placement (engine.DeviceTrades.place): the columns copied to `positions` - 1 places, 16 GiB apart, of ONE
    allocation, each probed with the bench's own step (device time by HIP events); the fastest stays.  Then the chosen copy is run until
    its level has settled (blocks of 5 steps, until one is not 0.5 % faster than the one before): the first passes over new memory run
    up to 10 % slower than the level they settle at (profiles/r04_step_timeline.txt).  Set-up, not a step.
    -> (the chosen trade set, {"probe_ms": [...], "offset_gib": [...], "chosen": k, "settle_kernel_ms": [...]})."""
    fns = {}

    def probe(t):
        if id(t) not in fns:
            fns[id(t)] = step_of(t)
        fns[id(t)]()

    chosen, info = trades.place(probe, positions=positions)
    fn = step_of(chosen)
    settle = [_probe_step_kernel_ms(ctx, fn, 5)]
    while len(settle) < 8:
        settle.append(_probe_step_kernel_ms(ctx, fn, 5))
        if settle[-1] > settle[-2] * 0.995:
            break
    info["settle_kernel_ms"] = settle
    info["policy"] = (f"finmlkit_amd.engine.DeviceTrades.place(probe, positions={positions}): the columns as allocated (position 0) and copied to "
                      f"{len(info['probe_ms']) - 1} places, 16 GiB apart, of one allocation; each probed with the step (HIP events), the fastest kept")
    return chosen, info


def _proc_start_ticks(pid):
    """Start time of a process (clock ticks since boot, /proc/<pid>/stat field 22): with the pid it names ONE process instance."""
    try:
        with open(f"/proc/{pid}/stat") as fh:
            return fh.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def rendezvous_path(world):
    """The same string on every rank of THIS job and on no other job: handed down by the self-spawning rank 0
    (FMK_BENCH_RDV), or -- under a launcher -- built from what the ranks share: the launcher's pid, its start time (a pid
    is reused, a (pid, start time) pair is not) and MASTER_PORT."""
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
    if os.environ.get("FMK_BENCH_RDV"):
        return os.environ["FMK_BENCH_RDV"]
    owner = os.getppid() if world > 1 else os.getpid()
    return os.path.join(base, f"fmk_comm_{os.environ.get('MASTER_PORT', '0')}_{owner}_{_proc_start_ticks(owner)}")


def spawn_ranks(args):
    """`--gpus N` (N > 1) with no launcher: this process is rank 0; ranks 1..N-1 are children running this same file with the
    same arguments.  -> list of Popen (rank 0 waits for them at the end and turns a failed child into its own rc)."""
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
    rdv = os.path.join(base, f"fmk_comm_self_{os.getpid()}_{_proc_start_ticks(os.getpid())}_{os.urandom(4).hex()}")
    common = {"WORLD_SIZE": str(args.gpus), "FMK_BENCH_RDV": rdv}
    os.environ.update(common, RANK="0", LOCAL_RANK="0")
    kids = []
    for r in range(1, args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r))
        # a child's stdout goes to OUR stderr: the one JSON line on stdout is rank 0's
        kids.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=sys.stderr))
    return kids


def reap(kids, grace_s=60.0):
    """Wait for the child ranks; -> worst return code (a child that does not finish in `grace_s` is killed: rc 124)."""
    worst = 0
    deadline = time.time() + grace_s
    for k in kids:
        try:
            rc = k.wait(timeout=max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            k.kill()
            k.wait()
            rc = 124
        worst = worst or rc
    return worst


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL's P2P between processes needs it on this driver
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")      # librccl's banner / warnings: not on stdout, that is the JSON line's
    kids = []
    with _StdoutIsTheJsonLine():
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            kids = spawn_ranks(args)                                 # (each is a `python bench.py` of its own and guards its own fd 1)
        try:
            rc = run(args)
        except BaseException:
            for k in kids:                                           # exactly the processes started above
                if k.poll() is None:
                    k.kill()
            raise
        rc = rc or reap(kids)
    sys.exit(rc)


class _StdoutIsTheJsonLine:
    """fd 1 carries the ONE JSON line and nothing else.  librccl prints its version banner with printf when the first communicator
    comes up (NCCL_DEBUG_FILE does not move it), and any other library may do the same: for the length of the run fd 1 points at
    stderr, and the line is written to a saved copy of the real stdout by `emit`."""
    saved = None

    def __enter__(self):
        sys.stdout.flush()
        try:
            saved = os.dup(1)
            try:
                os.dup2(2, 1)
            except OSError:                             # no usable stderr: leave stdout alone (the line is still its last one)
                os.close(saved)
                saved = None
        except OSError:
            saved = None
        _StdoutIsTheJsonLine.saved = saved
        return self

    @staticmethod
    def emit(text):
        fd = _StdoutIsTheJsonLine.saved
        if fd is None:                                  # (no guard, or no usable stderr: keep the line the LAST one of stdout)
            try:
                import ctypes as C
                C.CDLL(None).fflush(None)               # librccl's banner sits in the C stdio buffer
            except OSError:
                pass
            print(text, flush=True)
            return
        data = (text + "\n").encode()
        while data:
            data = data[os.write(fd, data):]

    def __exit__(self, *exc):
        try:
            import ctypes as C
            C.CDLL(None).fflush(None)                   # C stdio buffers of fd 1 (the banner) go where fd 1 points NOW: stderr
        except OSError:
            pass
        sys.stdout.flush()
        if _StdoutIsTheJsonLine.saved is not None:
            os.dup2(_StdoutIsTheJsonLine.saved, 1)
            os.close(_StdoutIsTheJsonLine.saved)
            _StdoutIsTheJsonLine.saved = None
        return False


class _Tee:
    """sys.stderr of a rank of a multi-rank run: everything also goes to the rank's own log file"""
    def __init__(self, a, b):
        self.a, self.b = a, b

    def write(self, x):
        self.a.write(x)
        self.b.write(x)
        self.b.flush()
        return len(x)

    def flush(self):
        self.a.flush()
        self.b.flush()


def rank_log(rank, world):
    """N > 1: every rank keeps its own record -- gpurun_out/bench_rank<r>.log (what this process writes to stderr: the transport it
    got, a fallback and its reason) and gpurun_out/bench_rank<r>.rccl.log (librccl's own NCCL_DEBUG output) -- next to the files the
    driver pulls, so that a run that fell back or died says why.  FMK_BENCH_LOGDIR overrides the directory."""
    if world <= 1:
        return
    d = os.environ.get("FMK_BENCH_LOGDIR") or os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        fh = open(os.path.join(d, f"bench_rank{rank}.log"), "w")
    except OSError:
        return
    sys.stderr = _Tee(sys.stderr, fh)
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    if os.environ.get("NCCL_DEBUG_FILE", "/dev/stderr") == "/dev/stderr":
        os.environ["NCCL_DEBUG_FILE"] = os.path.join(d, f"bench_rank{rank}.rccl.log")
    print(f"[bench] rank {rank} of {world}: pid {os.getpid()}, LOCAL_RANK {os.environ.get('LOCAL_RANK')}, "
          f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, args {sys.argv[1:]}", file=sys.stderr)


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    rank_log(rank, world)
    # developer switch for boxes with ONE GPU: every rank on device 0.  RCCL refuses a communicator with two ranks on one
    # device, so this switch ASKS for the host-staged transport (config.transport says so) -- it exercises the N > 1 flow end
    # to end (spawn, rendezvous, plan, exchange, boundary bar, gathers), not xGMI
    one_device = bool(os.environ.get("FMK_BENCH_ONE_DEVICE")) or bool(os.environ.get("FMK_BENCH_SAME_DEVICE_RCCL"))
    if os.environ.get("FMK_BENCH_ONE_DEVICE") and world > 1:
        args.transport = "host"
    # (FMK_BENCH_SAME_DEVICE_RCCL: every rank on device 0 but RCCL still asked for -- the test of the rc-3 fallback rule)
    os.environ["FMK_DEVICE"] = "0" if one_device else str(local_rank)

    comm = None
    use_dist = world > 1 or args.force_dist

    import numpy as np
    from finmlkit_amd import _ffi, engine
    from finmlkit_amd._ffi import DeviceArray, c_f64, c_i64
    import ctypes as C

    ctx = _ffi.default_context()
    selftest_deadline = float(os.environ.get("FMK_BENCH_SELFTEST_DEADLINE", "10"))      # 0: no selftest
    if world > 1 and selftest_deadline > 0:
        # first contact (finmlkit_amd/dist.py: selftest): the node as this rank sees it, a 1 KiB ncclSend / ncclRecv to rank + 1 with a
        # 10 s deadline, its content checked -- on record in gpurun_out/bench_rank<r>.log BEFORE anything is timed.  It decides
        # nothing: the communicator of the run is made (and falls back, with rc 3) below as before.
        from finmlkit_amd.dist import selftest
        try:
            selftest(rank, world, rendezvous_path(world) + ".selftest", ctx=ctx, log=sys.stderr,
                     deadline_s=selftest_deadline)
        except Exception as e:                                                 # noqa: BLE001 -- a report, never a reason to stop
            print(f"[selftest] rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
    n = args.ticks
    free, total = ctx.mem_info()
    need = n * 21 + (1 << 30)
    if need > free:
        n = int((free - (2 << 30)) // 21)
        if rank == 0:
            print(f"[bench] reducing ticks/GPU to {n} (free HBM {free / 2**30:.1f} GiB)", file=sys.stderr)
    trades = engine.DeviceTrades.synth(n, seed=args.seed, first=rank * n, ctx=ctx)
    ctx.sync()
    placement = None

    transport_note = None
    transport = "none"
    fell_back = False
    if use_dist:
        from finmlkit_amd.dist import Comm, ShardedTimeBars
        # the ranks of this node meet in a shared-memory file; rank 0 creates it, it is unlinked as soon as every rank
        # has attached
        path = rendezvous_path(world)
        transport = args.transport
        try:
            comm = Comm(ctx, rank, world, path, transport, self_loop=(world == 1))
            if world > 1:
                print(f"[bench] rank {rank}: communicator up, transport {transport}", file=sys.stderr)
        except _ffi.FmkError as e:
            if transport != "rccl":
                raise
            # RCCL could not initialise (every rank learns it in the rendezvous).  The same step over the host-staged
            # transport is correct but says nothing about xGMI: the run goes on so that the reason and the numbers are
            # on record (stderr), and ends with rc 3 -- a scaling curve must not silently be a host-staged one
            transport_note = f"host-staged fallback: {e}"
            fell_back = True
            print(f"[bench] rank {rank}: {transport_note}", file=sys.stderr)
            comm = Comm(ctx, rank, world, path + ".host", "host", self_loop=(world == 1))
            transport = "host"

    want_median = not args.no_median
    state = {}

    def clock_of(t0, t1):
        ne, e0, d = c_i64(), c_i64(), c_i64()
        _ffi.check(_ffi.lib().fmk_time_bar_clock(c_i64(t0), c_i64(t1), c_f64(args.interval), C.byref(ne),
                                                 C.byref(e0), C.byref(d)))
        return ne.value, e0.value, d.value

    def ensure_buffers(ne):
        if state.get("cap", 0) < ne:
            cap = ne + 1024
            state["cap"] = cap
            state["clock"] = DeviceArray(ctx, cap, np.int64)
            state["idx"] = DeviceArray(ctx, cap, np.int64)
            state["out"] = trades.alloc_ohlcv(cap, want_median)

    _pre = None
    if use_dist:
        # Before the placement probes: everything a sharded run allocates later -- RCCL's point-to-point channels (built by the first
        # exchange), the plan's index / output / boundary buffers -- is allocated once now.  Measured without this: the columns probed
        # at 2.13 ms and ran at 2.31 ms in the sharded step after those allocations (profiles/r04_sharded_step.txt).
        from finmlkit_amd.dist import ShardedTimeBars as _STB
        _pre = _STB(trades, rank, world, args.interval, want_median, self_loop=(world == 1)).setup(comm)
        _pre.step(comm)
        ctx.sync()
        comm.sync()
    def step_of(t):
        t0, t1 = t.first_last_ts()
        ne, e0, d = clock_of(t0, t1)
        ensure_buffers(ne)
        if args.separate_index:
            def fn():
                clock, ci = t.time_bar_index(args.interval, clock_params=(ne, e0, d), out=(state["clock"], state["idx"]))
                t.bar_ohlcv(ci, want_median=want_median, out=state["out"])
        else:
            def fn():
                t.time_bars_ohlcv(args.interval, want_median, clock_params=(ne, e0, d),
                                  out_index=(state["clock"], state["idx"]), out=state["out"])
        return fn

    if args.placements > 1:
        # opt-in diagnostic: WHERE the 21 GB of input columns land decides the level of the dominant kernel (+-5 % between allocations
        # of one process, constant for the life of an allocation: profiles/r04_placement.txt).  The default run does NOT do this: its
        # headline is what the library delivers on the columns as it allocated them.
        trades, placement = choose_placement(ctx, trades, args, rank, n, step_of, args.placements)

    if use_dist:
        # SET-UP, not a step: the plan (global clock, edge partition, halo lengths, boundary buffers) is a function of
        # the immutable columns -- two host all-gathers, once.  The first exchange also builds RCCL's point-to-point
        # channels (seconds, once).
        # (the set-up made before the placement probes goes first: its index / output / boundary buffers return to the allocator's free
        #  list and come back, same sizes, to the set-up below -- no allocation is made or released after the probes.  With both alive the
        #  run's kernel took 2.30 ms where the probes had said 2.13: end-of-round run, profiles/r04_sharded_step.txt)
        _pre = None
        import gc
        gc.collect()
        shard = ShardedTimeBars(trades, rank, world, args.interval, want_median, self_loop=(world == 1)).setup(comm)
        shard.step(comm)
        ctx.sync()
        comm.sync()

    def step():
        if use_dist:
            # one neighbour exchange (communicator's stream) || index + interior bars, then the boundary bar: enqueue
            # only -- no read-back, no host wait, no allocation
            state["n_bars"] = shard.step(comm)
            return state["n_bars"]
        t0, t1 = trades.first_last_ts()
        ne, e0, d = clock_of(t0, t1)
        ensure_buffers(ne)
        if args.separate_index:
            clock, ci = trades.time_bar_index(args.interval, clock_params=(ne, e0, d), out=(state["clock"], state["idx"]))
            trades.bar_ohlcv(ci, want_median=want_median, out=state["out"])
        else:
            # clock + close indices + comp_bar_ohlcv incl. the median trade size in ONE library call (fmk_time_bars_ohlcv_dev): the
            # indexer kernels, then the dominant kernel k_bar_ohlcv_small (round 4's in-kernel edge search measured slower and is
            # gone: profiles/r04_indexer.txt)
            trades.time_bars_ohlcv(args.interval, want_median, clock_params=(ne, e0, d),
                                   out_index=(state["clock"], state["idx"]), out=state["out"])
        state["n_bars"] = ne - 1
        return ne - 1

    def barrier():
        if comm:
            comm.sync()          # first: this wait has a deadline (a dead neighbour is an error here, not a hang in ctx.sync)
        ctx.sync()
        if comm:
            comm.barrier()

    for _ in range(args.warmup):
        step()
    ctx.call("fmk_profile_enable", C.c_int(1))      # HIP-event pair around every dominant-kernel launch
    if comm:
        comm.profile_enable(True)                   # ... and around every halo exchange, on the communicator's stream
    barrier()
    t_start = time.perf_counter()
    for k in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_start
    kms = (C.c_double * 256)()
    kn = C.c_int()
    ctx.call("fmk_profile_read", kms, C.c_int(256), C.byref(kn))
    total_launches = c_i64()
    ctx.call("fmk_profile_count", C.byref(total_launches))
    ctx.call("fmk_profile_enable", C.c_int(0))

    k_ms = [kms[i] for i in range(kn.value)]
    # the pipelined time-bar step launches the dominant kernel TWICE per step (the first eighth of the bars, then the rest): the
    # kernel time of a step is the sum of its launches, each timed on its own (the bubble between them is not kernel time)
    # launches per step from the library's own count (with steps x launches > 256 the ring of event pairs wraps; the
    # ring then holds the LAST 256 launches, oldest at slot total mod 256 -- put them in order and keep whole steps)
    total_launches = int(total_launches.value)
    lps = total_launches // args.steps if args.steps and total_launches % args.steps == 0 and total_launches >= args.steps else 1
    if total_launches > len(k_ms):
        start = total_launches % len(k_ms)
        k_ms = k_ms[start:] + k_ms[:start]
    if lps > 1:
        whole = len(k_ms) // lps
        k_ms = k_ms[len(k_ms) - whole * lps:]
        k_ms = [sum(k_ms[i * lps:(i + 1) * lps]) for i in range(whole)]
    avg_k_ms = sum(k_ms) / len(k_ms)
    per_rank = None
    if comm:
        x_ms = comm.profile_read()
        comm.profile_enable(False)
        # per rank: its own wall time of the K steps, the average of its dominant kernel, the average of its exchange
        rows = comm.all_gather_f64([elapsed, avg_k_ms, sum(x_ms) / len(x_ms) if x_ms else float("nan")])
        elapsed = max(r[0] for r in rows)                                 # MAX over ranks
        per_rank = {"ms_per_step": [r[0] / args.steps * 1e3 for r in rows],
                    "kernel_ms": [r[1] for r in rows],
                    "exchange_ms": [r[2] for r in rows]}
        nb_all = comm.all_gather_i64([state["n_bars"]])
        n_bars_total = sum(x[0] for x in nb_all)
    else:
        n_bars_total = state["n_bars"]

    nb = state["n_bars"]
    # algorithmic bytes of ONE launch of the dominant kernel: price f64 + amount f32 read once per tick,
    # close_idx read once and 60 (+8 with the median) B written per bar (DESIGN.md "roofline")
    alg_bytes = n * 12 + nb * (68 if want_median else 60) + (nb + 1) * 8
    achieved = alg_bytes / (avg_k_ms * 1e-3) / 1e9

    tc = _traffic_constants()
    tc_ok = bool(tc) and want_median and args.interval == 60.0 and tc.get("write_bytes_per_bar") is not None
    tc_stale = bool(tc_ok) and tc.get("kernel_source_sha256") != kernel_source_sha256()
    if rank == 0:
        total_ticks = n * world
        line = {
            "metric": "ticks/sec aggregated to bars",
            "value": total_ticks * args.steps / elapsed,
            "unit": "ticks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"cfg2: {n:.3g} synthetic ticks/GPU -> {args.interval:g}s time bars: clock+close indices, "
                            f"OHLC/volume/VWAP/trades{'' if args.no_median else ' + median trade size'}",
                "ticks_per_gpu": n, "n_bars_total": n_bars_total, "interval_s": args.interval,
                "parallelism": (f"time-range shards x{world}, 1 neighbour halo exchange per step "
                                f"({'ncclSend/ncclRecv of librccl' if transport == 'rccl' else 'host-staged'} behind the "
                                f"C ABI, no PyTorch{'; self-loop diagnostic' if world == 1 else ''})") if use_dist else "1 GPU",
                # ALWAYS present: "none" (one GPU, no exchange), "rccl" (ncclSend/ncclRecv over xGMI) or "host" (staged
                # through shared memory: asked for with --transport host / FMK_BENCH_ONE_DEVICE, or -- rc 3 -- a fallback)
                "transport": transport,
                "launcher": ("external launcher" if os.environ.get("TORCHELASTIC_RUN_ID") or not os.environ.get("FMK_BENCH_RDV") else "self-spawned ranks") if world > 1 else "none",
            },
            "roofline": {"bound": "hbm",
                         "kernel": "k_bar_ohlcv_small<f32 amount, exact 17..21-chunk classes, %s>" % ("fused median" if want_median else "no median"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # per launch, like `achieved` (a step's launches cover disjoint bar ranges: bytes of the step / launches)
                         "traffic": (tc["read_bytes_per_tick"] * n + tc["write_bytes_per_bar"] * nb) / lps if tc_ok else None,
                         # true when the kernel's sources have changed since the counters were read (SHA-256 in the constants)
                         "traffic_stale": tc_stale if tc_ok else None,
                         "traffic_source": (f"offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel (sources "
                                            f"sha256 {str(tc.get('kernel_source_sha256'))[:12]}, commit "
                                            f"{tc['commit']}; profiles/traffic_constants.json: FETCH_SIZE x{tc['fetch_size_correction']}, "
                                            f"WRITE_SIZE x{tc['write_size_correction']}, calibrated on known byte counts in the same passes), "
                                            f"scaled to this run's ticks and bars; not collected in this run") if tc_ok else None,
                         "algorithmic_bytes_per_launch": alg_bytes / lps, "avg_kernel_ms": avg_k_ms,
                         "avg_launch_ms": avg_k_ms / lps, "launches_per_step": lps,
                         "launches_note": (None if lps == 1 else
                                           f"{lps} launches of the kernel per step over disjoint bar ranges (pipelined time-bar step): "
                                           "achieved = bytes of a step / summed duration of its launches = bytes per launch / average "
                                           "launch duration (avg_launch_ms is what a rocprofv3 kernel summary averages)"),
                         "kernel_ms_min": min(k_ms), "kernel_ms_max": max(k_ms),
                         "launches_timed": len(k_ms) * lps},
        }
        line["roofline"]["columns"] = ("as the library allocated them (default)" if not placement else
                                       "placed by DeviceTrades.place() before the timed region (--placements: a diagnostic run)")
        if placement:
            line["roofline"]["placement"] = placement
        if per_rank:
            ms = per_rank["ms_per_step"]
            line["per_rank"] = {"ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "ms_per_step": ms,
                                "dominant_kernel_ms": per_rank["kernel_ms"],
                                "exchange_ms": per_rank["exchange_ms"],
                                "exchange_note": ("HIP events on the communicator's stream around the ncclGroup of each step "
                                                  "(device time of the send/recv alone; it overlaps the interior bars)"
                                                  if transport == "rccl" else "wall clock of the host-staged copy")}
        if transport_note:
            line["config"]["transport_note"] = transport_note
            line["valid"] = False
        if world == 1 and not use_dist and not placement and args.placed_probe > 1:
            # AUXILIARY, after the timed region and outside `value` / `frac`: the same step on columns placed by the library's opt-in
            # DeviceTrades.place() -- what a caller who asks for it gets (the headline above is the default: columns as allocated)
            try:
                placed, pinfo = choose_placement(ctx, trades, args, rank, n, step_of, args.placed_probe)
                fnp = step_of(placed)
                pk = _probe_step_kernel_ms(ctx, fnp, args.steps)
                ctx.sync()
                tp0 = time.perf_counter()
                for _ in range(args.steps):
                    fnp()
                ctx.sync()
                p_ms = (time.perf_counter() - tp0) / args.steps * 1e3
                fr_all = [alg_bytes / (m * 1e-3) / 1e9 / HBM_PEAK_GBS for m in pinfo["probe_ms"]]
                line["roofline"]["frac_as_allocated"] = line["roofline"]["frac"]
                line["roofline"]["placed"] = {
                    "entry_point": "finmlkit_amd.engine.DeviceTrades.place(probe, positions)", "frac": alg_bytes / (pk * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "kernel_ms": pk, "ms_per_step": p_ms, "probe_step_ms": pinfo["probe_ms"], "probe_offset_gib": pinfo["offset_gib"],
                    "chosen": pinfo["chosen"], "settle_kernel_ms": pinfo["settle_kernel_ms"],
                    "probe_frac_of_step_ms_min_max": [min(fr_all), max(fr_all)],
                    "note": "opt-in, set-up once per trade set (positions x one device-to-device copy of the columns + the probes); "
                            "NOT the headline: `frac` / `value` are measured on the columns as allocated"}
                del placed, fnp
            except Exception as e:                                   # noqa: BLE001 -- auxiliary
                line["roofline"]["placed"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1:
            line["cpu_baseline"] = cpu_baseline(args)
            if not args.no_extras:
                line["other_configs"] = other_configs(trades, ctx, args)
        # a fallback run is NOT a result: its line goes to stderr and the process ends with rc 3
        if fell_back:
            print(json.dumps(line), flush=True, file=sys.stderr)
        else:
            _StdoutIsTheJsonLine.emit(json.dumps(line))
    if comm:
        comm.barrier()
        comm.close()
    return 3 if fell_back else 0


if __name__ == "__main__":
    main()
