#!/usr/bin/env python3
"""Randomized campaign for _volume_bar_indexer on every tier of fmk_volume.hip (2048-tick LDS tables, global tables, chain
walk, serial walk) and its certification -- and, with kind = dollar, for _dollar_bar_indexer (closed form / serial walk):
exact mode == oracle always; fast mode may differ only with a reported decision.
usage: fuzz_volume.py [seed] [cases] [max_n] [volume|dollar]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from oracle import oracle as orc


def amounts(rng, n):
    kind = rng.choice(["lognormal64", "lognormal32", "decimal", "cents", "integer", "quarter", "uniform", "sparse"])
    if kind == "lognormal64":
        a = rng.lognormal(0.0, float(rng.choice([0.1, 1.0, 2.5])), n)
    elif kind == "lognormal32":
        a = rng.lognormal(0.0, float(rng.choice([0.3, 1.0, 2.0])), n).astype(np.float32)
    elif kind == "decimal":
        a = rng.integers(0 if rng.random() < 0.3 else 1, 10, n) / 10.0
        if rng.random() < 0.5:
            a = a.astype(np.float32)
    elif kind == "cents":
        a = rng.integers(1, 500, n) / 100.0
    elif kind == "integer":
        a = rng.integers(0 if rng.random() < 0.3 else 1, 20, n).astype(np.float64 if rng.random() < 0.5 else np.float32)
    elif kind == "quarter":
        a = (rng.integers(1, 9, n) * 0.25).astype(np.float32)
    elif kind == "uniform":
        a = rng.random(n)
    else:                                                           # mostly zeros
        a = np.where(rng.random(n) < 0.02, rng.lognormal(0, 1, n), 0.0)
    if rng.random() < 0.08 and n > 10:                              # a prefix that dwarfs the threshold: one ulp of it is more
        a = a.astype(np.float64); a[rng.integers(0, n, 1 + n // 200000)] *= float(rng.choice([1e9, 1e12]))   # than 1e-11 * thr
    if rng.random() < 0.2 and n > 10:                               # whales
        a = a.copy(); a[rng.integers(0, n, max(1, n // 50000 + 1))] *= float(rng.choice([1e3, 1e5]))
    if rng.random() < 0.04 and n > 10:                              # outside the parallel domain
        a = a.copy(); a[rng.integers(0, n)] = rng.choice([-1.0, np.nan])
    return kind, a


def threshold(rng, a):
    n = len(a)
    a64 = np.abs(np.nan_to_num(a.astype(np.float64)))
    mean = float(a64.mean()) or 1.0
    r = rng.random()
    if r < 0.55:
        L = float(np.exp(rng.uniform(np.log(0.5), np.log(min(3e5, 4.0 * n)))))
        return mean * L
    if r < 0.75:
        return float(rng.choice([1.0, 10.0, 25.0, 100.0, 1000.0, 2500.0, 4000.0, 10000.0, 65536.0]))
    if r < 0.93:                                                    # some tick's sequential running sum, to the last bit
        cs = np.cumsum(a.astype(np.float64))
        v = float(cs[int(rng.integers(0, n))])
        return v if v > 0 and np.isfinite(v) else mean * 100
    return float(a64.sum()) * float(rng.choice([1.0, 1.0 + 1e-15, 2.0]))   # the total, a hair above it, unreachable


def prices(rng, n):
    p = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=n)), 0.01)
    if rng.random() < 0.3:
        p = np.round(p, 1)                                          # a coarse grid: many equal products
    if rng.random() < 0.04 and n > 10:
        p = p.copy(); p[rng.integers(0, n)] = rng.choice([np.nan, -5.0])
    return p


def run(seed, cases, max_n, kind="volume", verbose=True):
    """-> (failures, reported, fast_diff)"""
    rng = np.random.default_rng(seed)
    ctx = _ffi.default_context()
    bad = fast_diff = reported = 0
    for case in range(cases):
        n = int(np.exp(rng.uniform(np.log(1), np.log(max_n))))
        if rng.random() < 0.15:
            n = int(rng.choice([1, 2, 511, 512, 513, 2047, 2048, 2049, 4096, 4097, 65536, 65537, 524288, 524289]))
            n = min(n, max_n)
        dist, a = amounts(rng, n)
        if kind == "dollar":
            px = prices(rng, n)
            thr = threshold(rng, a.astype(np.float64) * px)
            want = orc._dollar_bar_indexer(px, a, thr)
        else:
            px = np.ones(n)
            thr = threshold(rng, a)
            want = orc._volume_bar_indexer(a, thr)
        t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), px, a)
        index = t.dollar_bar_index if kind == "dollar" else t.volume_bar_index
        tag = f"{kind} seed {seed} case {case}: {dist} {a.dtype} n={n} thr={thr!r} ({len(want) - 1} bars)"
        try:
            exact = index(thr).to_host()
            unc_exact = t.last_uncertified
            ctx.set_fast_threshold(True)
            try:
                fast = index(thr).to_host()
                unc = t.last_uncertified
            finally:
                ctx.set_fast_threshold(False)
        except Exception as e:                                      # noqa: BLE001
            print("ERROR", tag, type(e).__name__, e); bad += 1; continue
        if not np.array_equal(exact, want) or unc_exact != 0:
            k = int(np.argmax(exact[:min(len(exact), len(want))] != want[:min(len(exact), len(want))])) if len(exact) and len(want) else 0
            print("MISMATCH (exact mode)", tag, "uncertified", unc_exact, "lens", len(exact), len(want), "at", k,
                  exact[max(0, k - 1):k + 2], want[max(0, k - 1):k + 2]); bad += 1
        reported += unc > 0
        if not np.array_equal(fast, want):
            fast_diff += 1
            if unc == 0:
                print("UNREPORTED (fast mode)", tag); bad += 1
    if verbose:
        print(f"{kind} seed {seed}: {cases} cases, {bad} failures; fast mode reported decisions in {reported} cases and differed "
              f"from the reference in {fast_diff} (all reported)" if not bad else f"{kind} seed {seed}: {bad} FAILURES of {cases}")
    return bad, reported, fast_diff


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    max_n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 3_000_000
    kind = sys.argv[4] if len(sys.argv) > 4 else "volume"
    sys.exit(1 if run(seed, cases, max_n, kind)[0] else 0)


if __name__ == "__main__":
    main()
