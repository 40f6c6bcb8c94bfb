#!/usr/bin/env python3
"""Randomized campaign for _cusum_bar_indexer (reference logic.py:152-221) on streams the general fuzzer's tape does not draw: up to
millions of ticks (what the chain walk of fmk_cusum_chain.hip is chosen for), geometric and grid price paths with jumps and flat
stretches, sigma with NaN prefixes / holes / zeros, thresholds that some window's cumulative log return reaches exactly.
HIP path vs oracle: close indices and the in-place forward fill of sigma, bit for bit.  Run it as is and with the tier knobs of
DESIGN.md 4 (FMK_CUSUM_CHAIN=2 FMK_CUSUM_CHAIN_MIN_CHUNKS=2: the chain walk for every call).
usage: fuzz_cusum.py [cases] [seed] [max_n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finmlkit_amd.bar import logic
from finmlkit_amd.feature.core import utils as futils
from oracle import oracle as orc


def stream(rng, n):
    gap = int(rng.choice([1, 1000, 10**6]))
    rep = float(rng.choice([0.0, 0.3, 0.9]))                        # share of ticks that repeat the timestamp before them
    d = rng.integers(1, 3 * gap + 1, size=n)
    d[rng.random(n) < rep] = 0
    ts = 1_700_000_000_000_000_000 + np.cumsum(d).astype(np.int64)
    kind = rng.choice(["grid", "geometric", "jumps", "flat", "small"])
    vol = float(rng.choice([1e-5, 1e-4, 1e-3]))
    if kind == "grid":
        step = float(rng.choice([0.5, 0.01]))
        px = np.maximum(100.0 + step * np.cumsum(rng.integers(-2, 3, size=n)), step)
    elif kind == "geometric":
        px = 100.0 * np.exp(np.cumsum(rng.normal(0.0, vol, size=n)))
    elif kind == "jumps":
        r = rng.normal(0.0, vol, size=n)
        r[rng.integers(0, n, max(1, n // 5000))] += rng.choice([-0.05, 0.05, 0.2], size=max(1, n // 5000))
        px = 100.0 * np.exp(np.cumsum(r))
    elif kind == "flat":
        r = rng.normal(0.0, vol, size=n)
        r[rng.random(n) < 0.95] = 0.0
        px = 100.0 * np.exp(np.cumsum(r))
    else:
        px = np.maximum(0.05 + 0.01 * np.cumsum(rng.integers(-1, 2, size=n)), 0.01)
    sk = rng.choice(["noisy", "const", "zeros", "steps"])
    if sk == "noisy":
        sig = np.abs(rng.normal(vol * 30, vol * 15, size=n))
    elif sk == "const":
        sig = np.full(n, vol * float(rng.choice([3, 30, 300])))
    elif sk == "zeros":
        sig = np.where(rng.random(n) < 0.5, 0.0, vol * 30)
    else:
        sig = np.repeat(np.abs(rng.normal(vol * 30, vol * 20, size=n // 997 + 1)), 997)[:n]
    if rng.random() < 0.5:
        sig[: int(rng.integers(0, min(n, 5000)))] = np.nan
    if rng.random() < 0.3:
        sig[rng.integers(0, n, size=max(1, n // 30))] = np.nan
    if rng.random() < 0.03:
        sig[:] = np.nan
    mult = float(rng.choice([0.5, 2.0, 5.0]))
    fl = vol * float(rng.choice([1, 10, 50, 500]))
    if rng.random() < 0.25 and n > 8:                                # a floor some window's log return reaches to the last bit
        a = int(rng.integers(1, n - 2))
        b = int(rng.integers(a + 1, min(n, a + int(rng.choice([3, 50, 3000]))) + 1))
        s = 0.0
        for x in np.log(px[a:b] / px[a - 1:b - 1]):
            s += float(x)
        if np.isfinite(s) and s != 0.0:
            fl = float(rng.choice([abs(s), np.nextafter(abs(s), np.inf), np.nextafter(abs(s), 0.0)]))
    return f"{kind}/{sk} rep={rep}", ts, px.astype(np.float64), sig, fl, mult


def replay(ts, r, sig, fl, mult):
    """logic.py:152-221 on given tick returns r[i] (i >= 1) and the forward-filled sigma"""
    n = len(r)
    first = int(np.argmax(~np.isnan(sig))) if (~np.isnan(sig)).any() else 0
    out = [first]
    sp = sn = 0.0
    rl, sl, tl = r.tolist(), sig.tolist(), ts.tolist()
    i = first + 1
    while i < n:
        sp = max(0.0, sp + rl[i])
        sn = min(0.0, sn + rl[i])
        if i + 1 < n and tl[i] == tl[i + 1]:
            i += 1
            continue
        lam = mult * sl[i]
        if lam != lam:                                              # sigma all NaN: no comparison holds
            i += 1
            continue
        if not lam > fl:
            lam = fl
        if sp >= lam:
            out.append(i); sp = 0.0
        elif sn <= -lam:
            out.append(i); sn = 0.0
        i += 1
    return out


def _onepass():
    import ctypes as C
    from finmlkit_amd import _ffi
    v = [C.c_int64() for _ in range(4)]
    _ffi.lib().fmk_diag_cusum_onepass(*(C.byref(x) for x in v))
    return tuple(x.value for x in v)


def _tier(n):
    """which tier answered the last call (the chain walk reports itself through fmk_diag_cusum_last)"""
    import ctypes as C
    from finmlkit_amd import _ffi
    if n < 2:
        return "none (n < 2 or an exception)"
    t = C.c_int64()
    _ffi.lib().fmk_diag_cusum_last(C.byref(t), None, None)
    if _onepass()[0] == 1:
        return "one pass"
    return "chain walk" if t.value == 1 else "fixed point"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    max_n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 2_000_000
    rng = np.random.default_rng(seed)
    bad = ulp = 0
    tiers = {"chain walk": 0, "one pass": 0, "fixed point": 0, "none (n < 2 or an exception)": 0}
    one_pass_chunks = 0
    for case in range(cases):
        n = int(np.exp(rng.uniform(np.log(1), np.log(max_n))))
        if rng.random() < 0.15:
            n = min(max_n, int(rng.choice([1, 2, 3, 511, 512, 513, 2047, 2048, 2049, 65536, 65537, 131073])))
        what, ts, px, sig, fl, mult = stream(rng, n)
        tag = f"cusum seed {seed} case {case}: {what} n={n} floor={fl!r} mult={mult}"
        s_want, s_got = sig.copy(), sig.copy()
        try:
            want, s_want = orc._cusum_bar_indexer(ts, px, s_want, fl, mult, return_sigma=True)
            w_exc = None
        except Exception as e:                                      # noqa: BLE001
            want, w_exc = None, type(e).__name__
        try:
            got = logic._cusum_bar_indexer(ts, px, s_got, fl, mult)
            g_exc = None
        except Exception as e:                                      # noqa: BLE001
            got, g_exc = None, type(e).__name__
        tiers[_tier(n if g_exc is None else 0)] += 1
        if g_exc is None and _onepass()[0] == 1:
            one_pass_chunks += _onepass()[3]
        if w_exc or g_exc:
            if w_exc != g_exc:
                print("EXCEPTION", tag, "oracle", w_exc, "hip", g_exc); bad += 1
            continue
        if not np.array_equal(got, want):
            # tick returns are within 1 ulp of the reference's np.log(p / pm), not the same double -- a close may move
            # when a cumulative sum lands on its threshold to the last bit, which is what the knife-edge floors above arrange (and
            # price grids with few distinct quotients make common).  Such a case is counted apart when the reference's loop, run
            # here on the DEVICE's tick returns (comp_lagged_returns over one-second timestamps: the same fmk_log_ratio), gives
            # the device's closes and those returns are within one ulp of np.log's.
            r_dev = futils.comp_lagged_returns(np.arange(n, dtype=np.int64) * 1_000_000_000, px, 1.0, True)
            with np.errstate(all="ignore"):
                r_np = np.log(px[1:] / px[:-1])
            one_ulp = bool(np.all(np.abs(r_dev[1:] - r_np) <= np.spacing(np.abs(r_np))))
            if one_ulp and replay(ts, r_dev, s_want, fl, mult) == list(got):
                ulp += 1
                continue
            m = min(len(got), len(want))
            k = int(np.argmax(got[:m] != want[:m])) if (got[:m] != want[:m]).any() else m
            print("MISMATCH", tag, "lens", len(got), len(want), "at", k, got[max(0, k - 1):k + 2], want[max(0, k - 1):k + 2]); bad += 1
        elif not np.array_equal(s_got, s_want, equal_nan=True):
            print("SIGMA FILL", tag); bad += 1
    print("answered by:", ", ".join(f"{k} {v}" for k, v in tiers.items()), f"; chunks of 4096 ticks walked by the one-pass form: {one_pass_chunks}")
    print(f"{cases} cusum cases, seed {seed}, sizes up to {max_n}: {bad} failures; {ulp} cases where the last bit of a logarithm "
          f"decides a close (the reference's loop on the device's tick returns, all within one ulp of np.log's, gives the device's closes)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
