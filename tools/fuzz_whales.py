#!/usr/bin/env python3
"""_dollar_bar_indexer on tapes with BLOCK TRADES (increments >= the threshold: the backlog of closes they leave is what used to send
the whole stream to the serial walk): random streams with a share of trades 0.5 .. 60 thresholds large, exact mode against the sequential
oracle, n_uncertified must come back 0.  usage: fuzz_whales.py [seed] [cases] [max_n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd import _ffi, engine
from oracle import oracle as orc


def case(rng, max_n):
    n = int(np.exp(rng.uniform(np.log(200), np.log(max_n))))
    kind = rng.choice(["lognormal32", "dyadic", "decimal32", "uniform64"])
    if kind == "lognormal32":
        a = rng.lognormal(0.0, float(rng.choice([0.3, 1.0])), n).astype(np.float32)
    elif kind == "dyadic":
        a = (rng.integers(1, 4097, n) / 1024.0).astype(np.float32)
    elif kind == "decimal32":
        a = (rng.integers(1, 100, n) / 10.0).astype(np.float32)
    else:
        a = rng.random(n) + 0.01
    px = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=n)), 0.01)
    if rng.random() < 0.3:
        px = np.round(px, 1)
    L = float(np.exp(rng.uniform(np.log(3), np.log(min(3000, max(4, n / 4))))))
    d = a.astype(np.float64) * px
    thr = float(d.mean()) * L
    share = float(rng.choice([1e-4, 1e-3, 1e-2, 5e-2]))
    k = max(1, int(n * share))
    idx = rng.integers(0, n, k)
    size = np.exp(rng.uniform(np.log(0.5), np.log(60.0), k)) * thr / px[idx]        # 0.5 .. 60 thresholds each
    a = a.astype(np.float64) if a.dtype == np.float64 else a.copy()
    a[idx] = size.astype(a.dtype)
    if rng.random() < 0.3:                                                           # clusters of block trades (backlogs that overlap)
        j = int(rng.integers(0, max(1, n - 8)))
        a[j:j + 6] = (np.exp(rng.uniform(np.log(1.0), np.log(20.0), min(6, n - j))) * thr / px[j:j + 6]).astype(a.dtype)
    if rng.random() < 0.25:
        thr = float(2.0 ** np.round(np.log2(thr)))                                   # a power of two: binade edges on the threshold
    return kind, px, a, thr


def run(seed, cases, max_n, verbose=True):
    """FMK_DL_FORCE_EXACT_TIER=1 (set by main) sends every case through the exact tier; fmk_diag_dollar_last says which path answered
    each call (1 / 2: the exact tier served the stream, 3: it handed the call to the serial walk)."""
    import ctypes as C
    rng = np.random.default_rng(seed)
    bad = 0
    served = tried = 0
    for c in range(cases):
        kind, px, a, thr = case(rng, max_n)
        n = len(a)
        want = orc._dollar_bar_indexer(px, a, thr)
        t = engine.DeviceTrades.from_numpy(np.arange(n, dtype=np.int64), px, a)
        got = t.dollar_bar_index(thr).to_host()
        unc = t.last_uncertified
        path = C.c_int64(-1)
        _ffi.lib().fmk_diag_dollar_last(C.byref(path))
        tried += path.value in (1, 2, 3)
        served += path.value in (1, 2)
        if unc != 0 or not np.array_equal(got, want):
            m = min(len(got), len(want))
            k = int(np.argmax(got[:m] != want[:m])) if m and (got[:m] != want[:m]).any() else m
            print(f"MISMATCH whales seed {seed} case {c}: {kind} {a.dtype} n={n} thr={thr!r} bars {len(want) - 1} uncertified {unc} "
                  f"lens {len(got)} {len(want)} first difference at {k}: {got[max(0, k - 1):k + 2]} vs {want[max(0, k - 1):k + 2]}")
            bad += 1
    if verbose:
        print(f"whales seed {seed}: {cases} cases, {bad} failures; exact tier tried {tried} times, served {served} "
              f"(the others: a backlog beyond 500 thresholds or a replay that did not re-join -> serial walk)")
    return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    max_n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 2_000_000
    os.environ.setdefault("FMK_DL_FORCE_EXACT_TIER", "1")
    sys.exit(1 if run(seed, cases, max_n) else 0)
