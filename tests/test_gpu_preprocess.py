"""GPU parity of the TradesData(preprocess=True) loops ("next" rank 4): merge_split_trades, comp_trade_side_vector and
the whole pipeline against reference-generated goldens and the CPU oracle."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu


def _stream(seed, n, second_resolution=False, flat=0.6):
    rng = np.random.default_rng(seed)
    burst = rng.geometric(0.45, n)
    ts = np.repeat(1_700_000_000_000_000_000 + np.cumsum(rng.integers(1, 50_000_000, n)), burst)[:n]
    if second_resolution:
        ts = ts // 1_000_000_000 * 1_000_000_000
    px = (2_700_000 + np.cumsum(rng.choice([-1, 0, 0, 0, 1], n))) * 0.01
    am = ((1 + rng.integers(0, 4096, n)) * 2.0 ** -10).astype(np.float32)
    am[rng.random(n) < 0.3] = np.float32(0.1)
    ibm = rng.random(n) < 0.5
    ibm[1:][rng.random(n - 1) < flat] = False
    return ts.astype(np.int64), px, am, ibm


def _check_merge(got, want, what):
    for g, w, k in zip(got, want, ("ts", "px", "am", "side")):
        assert g.dtype == w.dtype, (what, k)
        np.testing.assert_array_equal(g, w, err_msg=f"{what}:{k}")


def test_merge_and_tick_rule_golden():
    from finmlkit_amd.bar.utils import comp_trade_side_vector, merge_split_trades
    d = G.load("preprocess")
    for name in ("a", "b", "c", "eps"):
        ts, px, am, ibm = (d[f"{name}__{k}"] for k in ("ts", "px", "am", "ibm"))
        for tag, flag in (("side", ibm), ("noside", None)):
            if f"{name}__{tag}_ts" not in d:
                continue
            _check_merge(merge_split_trades(ts, px, am, flag),
                         tuple(d[f"{name}__{tag}_{k}"] for k in ("ts", "px", "am", "sd")), f"{name} {tag}")
        np.testing.assert_array_equal(comp_trade_side_vector(px), d[f"{name}__tickrule"])


@pytest.mark.parametrize("n,sec,with_maker", [(1_000_000, False, True), (1_000_000, True, True), (300_001, False, False),
                                             (4097, True, False), (1, False, True), (2, False, True)])
def test_merge_vs_oracle(orc, n, sec, with_maker):
    from finmlkit_amd.bar.utils import merge_split_trades
    ts, px, am, ibm = _stream(n, n, sec)
    flag = ibm if with_maker else None
    _check_merge(merge_split_trades(ts, px, am, flag), orc.merge_split_trades(ts, px, am, flag), f"n={n}")


def test_merge_one_long_run(orc):
    """Degenerate input: every trade shares the timestamp and maker flag (one coarse run walked by one thread)."""
    from finmlkit_amd.bar.utils import merge_split_trades
    n = 200_000
    ts = np.full(n, 1_700_000_000_000_000_000, dtype=np.int64)
    px = 100.0 + (np.arange(n) // 7) * 0.01
    am = np.full(n, 0.1, dtype=np.float32)
    ibm = np.zeros(n, dtype=bool)
    got = merge_split_trades(ts, px, am, ibm)
    _check_merge(got, orc.merge_split_trades(ts, px, am, ibm), "one run")
    assert len(got[0]) == (n + 6) // 7


@pytest.mark.parametrize("n", [1, 2, 2048, 2049, 1_000_003])
def test_tick_rule_vs_oracle(orc, n):
    from finmlkit_amd.bar.utils import comp_trade_side_vector
    rng = np.random.default_rng(n)
    px = 100.0 + np.cumsum(rng.choice([-0.01, 0.0, 0.0, 0.01], n))
    if n > 10_000:
        px[5_000:12_000] = px[5_000]                 # a flat stretch longer than several 2048-tick tiles
        px[20_000:20_010] += 5e-13                   # moves below the 1e-12 epsilon do not count
        px[30_000] = np.nan
    np.testing.assert_array_equal(comp_trade_side_vector(px), orc.comp_trade_side_vector(px))


def test_tradesdata_preprocess_golden():
    """TradesData(preprocess=True) end to end == the reference's own object (ms timestamps, shuffled rows, duplicated
    ids, an id gap across two minutes; maker flags / tick rule; proc_res rounding)."""
    from finmlkit_amd.bar.data_model import TradesData
    d = G.load("tradesdata")
    for name in ("mk", "tr", "sec"):
        maker = d[f"{name}__raw_maker"].copy() if f"{name}__raw_maker" in d else None
        proc_res = str(d[f"{name}__proc_res"]) or None
        t = TradesData(d[f"{name}__raw_ts"].copy(), d[f"{name}__raw_px"].copy(), d[f"{name}__raw_qty"].copy(),
                       d[f"{name}__raw_id"].copy(), is_buyer_maker=maker, preprocess=True, proc_res=proc_res)
        assert list(t.data.columns) == ["timestamp", "price", "amount", "side"]
        for col in ("timestamp", "price", "amount", "side"):
            got, want = t.data[col].values, d[f"{name}__out_{col}"]
            assert got.dtype == want.dtype, (name, col, got.dtype, want.dtype)
            np.testing.assert_array_equal(got, want, err_msg=f"{name}:{col}")
        assert t.data_ok == bool(d[f"{name}__data_ok"]) and len(t.discontinuities) == int(d[f"{name}__n_disc"])
        assert t.missing_pct == pytest.approx(float(d[f"{name}__missing_pct"]))
        assert t.orig_timestamp_unit == str(d[f"{name}__unit"])
        assert t.data.index.name == "datetime" and t.data.index[0] == __import__("pandas").to_datetime(t.data["timestamp"].iloc[0])
