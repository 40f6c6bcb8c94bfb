#!/usr/bin/env python3
"""Seeded differential test of cfg 4's ONE call (bars_fused: OHLCV + median, order flow, footprints) against the oracle and against the separate
reducers, over the three schedules that run kernels on the auxiliary stream beside the context's own (round 4): bars of about equal length
(medians beside the lane kernel), bars of lognormal length (sorted lanes; comp_bar_ohlcv's size classes beside them), long bars (hourly / daily
style: comp_bar_ohlcv beside the order-flow features).  FMK_FLOW_LANES=2 forces the lane schedules at these sizes.
    python tools/fuzz_fused.py [cases] [seed]      exit code 1 on any difference"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FMK_FLOW_LANES", "2")
import numpy as np


def case(rng, orc, k):
    from tests.test_gpu_fused import _check_all
    kind = ("equal", "lognormal", "long")[k % 3]
    if kind == "equal":
        mean = int(rng.choice([700, 900, 1200, 1800]))
        n = int(rng.integers(400_000, 1_600_000))
        lens = np.maximum(1, rng.normal(mean, mean * 0.05, int(n / mean * 1.2)).astype(np.int64))
    elif kind == "lognormal":
        mean = int(rng.choice([700, 900, 1400]))
        sigma = float(rng.choice([0.5, 1.0, 1.3]))
        n = int(rng.integers(600_000, 2_000_000))
        lens = np.maximum(1, rng.lognormal(np.log(mean) - sigma * sigma / 2, sigma, int(n / mean * 1.5)).astype(np.int64))
        lens[rng.integers(0, len(lens), 3)] = rng.integers(8193, 30000, 3)
    else:
        mean = int(rng.choice([9000, 20000, 70000]))
        n = int(rng.integers(800_000, 2_500_000))
        lens = np.maximum(1, rng.normal(mean, mean * 0.3, int(n / mean * 1.5) + 4).astype(np.int64))
    if rng.random() < 0.5:
        lens[rng.integers(0, len(lens), 4)] = 0                            # a few empty bars
    ci = np.concatenate([[-1], np.cumsum(lens) - 1])
    ci = ci[ci <= n - 1].astype(np.int64)
    if len(ci) < 3:
        return kind
    step = float(rng.choice([0.01, 0.05]))
    px = np.round(100.0 + np.cumsum(rng.integers(-1, 2, n)) * step, 2)
    sd = rng.choice(np.array([-1, 1, 1, -1, 0], np.int8), n) if rng.random() < 0.3 else rng.choice(np.array([-1, 1], np.int8), n)
    am = (rng.integers(1, 4097, n) / 1024.0).astype(np.float32) if rng.random() < 0.5 else rng.lognormal(-1, 1.2, n).astype(np.float32)
    _check_all(orc, px, am, sd, ci, f"case {k} ({kind}, mean {mean}, n {n}, {'dyadic' if am[0] * 1024 == int(am[0] * 1024) else 'lognormal'} sizes)", tick=step)
    return kind


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from oracle import oracle as orc
    orc.build()
    rng = np.random.default_rng(seed)
    fails = 0
    for k in range(cases):
        try:
            case(rng, orc, k)
        except Exception as e:      # noqa: BLE001
            fails += 1
            print(f"FAIL seed {seed} {str(e)[:1200]}", flush=True)
    print(f"{cases} fused cases (equal / lognormal / long bars in turn), seed {seed}: {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
