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


def lognormal_tape(d):
    """(amounts float64, prices) of the lognormal tape of cfg1_reference_timebars.npz (oracle/gen_cfg1.py: same draws)."""
    rng = np.random.default_rng(int(d["cfg3_logn_seed"]))
    m = int(d["cfg3_logn_n"])
    lam = rng.lognormal(-1.0, 1.2, m)
    lpx = np.maximum(100.0 + 0.01 * np.cumsum(rng.integers(-2, 3, size=m)), 0.01)
    return lam, lpx
