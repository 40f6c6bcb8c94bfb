"""Helpers to read tests/golden/*.npz (written by oracle/gen_golden.py from the reference)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

OHLCV_KEYS = ["open", "high", "low", "close", "volume", "vwap", "trades", "median"]
DIR_KEYS = ["ticks_buy", "ticks_sell", "volume_buy", "volume_sell", "dollars_buy", "dollars_sell",
            "mean_spread", "max_spread", "cum_ticks_min", "cum_ticks_max", "cum_volumes_min",
            "cum_volumes_max", "cum_dollars_min", "cum_dollars_max"]
FP_LIST_KEYS = ["price_levels", "buy_volumes", "sell_volumes", "buy_ticks", "sell_ticks",
                "buy_imbalances", "sell_imbalances"]
FP_BAR_KEYS = ["buy_imbalances_sum", "sell_imbalances_sum", "cot_price_levels",
               "imb_max_run_signed", "vp_skew", "vp_gini"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def cases(d):
    """Case names of a fixture whose keys look like '<case>__<field>'."""
    out = []
    for k in d:
        if "__" in k:
            c = k.split("__")[0]
            if c not in out:
                out.append(c)
    return out


def synth_from(orc, spec):
    seed, first, n, gap = (int(x) for x in spec)
    return orc.synth(seed, first, n, gap)


def reducer_stream(orc, d, case):
    """(prices, amounts, sides) of a reducers.npz case."""
    if case.startswith("syn_"):
        _, px, am, sd = synth_from(orc, d["syn__synth"])
    elif case.startswith("sparse_"):
        _, px, am, sd = synth_from(orc, d["sparse__synth"])
    else:
        px, am, sd = d["rnd__px"], d["rnd__am"], d["rnd__sd"]
    return px, am, sd


def assert_f64_close(got, want, rtol=1e-9, what=""):
    """The north-star float bar: <=1e-9 relative (NaN/inf positions must coincide)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    np.testing.assert_allclose(got, want, rtol=rtol, atol=0.0, equal_nan=True, err_msg=what)


def assert_f32_close(got, want, what="", max_ulp=1, max_frac=5e-3):
    """float32 outputs of the reducers are float64 sums rounded ONCE to float32.  The HIP path sums
    in a different (tree) order, so the float64 values agree to ~1e-15 relative, far inside the 1e-9
    north-star tolerance -- but the final float32 rounding can land on the other side when the
    float64 value sits within ~1e-12 of a float32 tie.  On the synthetic stream that is not
    vanishingly rare: prices on a 0.01 grid times dyadic amounts make ~0.1-0.5 % of the per-bar
    dollar sums EXACT float32 ties (see DESIGN.md "float32 outputs").  Hence: never more than 1 ulp,
    and on at most `max_frac` of the bars (at least one bar is always tolerated)."""
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    assert got.shape == want.shape, f"{what}: shape"
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), f"{what}: NaN pattern"
    g = got[~nan_g].view(np.int32).astype(np.int64)
    w = want[~nan_w].view(np.int32).astype(np.int64)
    diff = np.abs(g - w)
    assert diff.max(initial=0) <= max_ulp, f"{what}: {diff.max()} ulp"
    allowed = int(np.ceil(max_frac * len(g))) if len(g) else 0
    assert (diff > 0).sum() <= allowed, f"{what}: {(diff > 0).sum()} flips of {len(g)} (allowed {allowed})"
