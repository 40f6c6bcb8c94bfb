"""CPU-only tests of the host-side mirror of the reference interface (no compute calls)."""
import numpy as np
import pandas as pd
import pytest

from tests import _golden as G


def _trades(n=100, with_side=True):
    from finmlkit_amd.bar.data_model import TradesData
    ts = 1_700_000_000_000_000_000 + np.arange(n, dtype=np.int64) * 1_000_000_000
    px = 100 + np.arange(n) * 0.01
    qty = np.ones(n, dtype=np.float32)
    return TradesData(ts, px, qty, np.arange(n), side=np.where(np.arange(n) % 2, 1, -1).astype(np.int8)
                      if with_side else None)


def test_tradesdata_schema_and_view_range():
    t = _trades()
    df = t.data
    assert list(df.columns) == ["timestamp", "price", "amount", "id", "side"]
    assert df.index.name == "datetime" and isinstance(df.index, pd.DatetimeIndex)
    assert df["timestamp"].dtype == np.int64 and df["amount"].dtype == np.float32
    assert t.orig_timestamp_unit == "ns"
    t.set_view_range("2023-11-14 22:13:30", "2023-11-14 22:13:40")
    assert len(t.data) == 11 and t.start_date == pd.Timestamp("2023-11-14 22:13:30")
    with pytest.raises(ValueError):
        t.set_view_range("2023-11-15", "2023-11-14")


def test_tradesdata_type_errors():
    from finmlkit_amd.bar.data_model import TradesData
    a = np.arange(3)
    with pytest.raises(TypeError, match="ts must be"):
        TradesData([1, 2, 3], a.astype(float), a.astype(float))
    with pytest.raises(TypeError, match="px must be"):
        TradesData(a, [1.0], a.astype(float))
    with pytest.raises(TypeError, match="side must be"):
        TradesData(a, a.astype(float), a.astype(float), side=[1, 1, 1])
    with pytest.raises(ValueError, match="id is required"):
        TradesData(a, a.astype(float), a.astype(float), preprocess=True)


def test_timestamp_unit_inference():
    from finmlkit_amd.bar.data_model import TradesData
    a = np.arange(3)
    for base, unit in ((1_700_000_000, "s"), (1_700_000_000_000, "ms"), (1_700_000_000_000_000, "us")):
        assert TradesData(a + base, a.astype(float), a.astype(float)).orig_timestamp_unit == unit


def test_price_tick_size_golden():
    from finmlkit_amd.bar.utils import comp_price_tick_size
    from oracle import oracle as orc
    d = G.load("tick_size")
    _, px, _, _ = G.synth_from(orc, d["synth"])
    assert comp_price_tick_size(px) == float(d["synth_tick"])
    for c in G.cases(d):
        assert comp_price_tick_size(d[f"{c}__px"]) == float(d[f"{c}__tick"]), c
    with pytest.raises(ValueError, match="Empty"):
        comp_price_tick_size(np.array([]))


def _fake_csr():
    off = np.array([0, 2, 5, 6], dtype=np.int64)
    flat = dict(price_levels=np.array([10, 11, 10, 11, 12, 13], np.int32),
                buy_volumes=np.arange(6, dtype=np.float32), sell_volumes=np.arange(6, dtype=np.float32)[::-1].copy(),
                buy_ticks=np.ones(6, np.int32), sell_ticks=np.full(6, 2, np.int32),
                buy_imbalances=np.array([0, 1, 0, 0, 1, 0], np.uint8), sell_imbalances=np.zeros(6, np.uint8))
    bar = dict(buy_imbalances_sum=np.array([1, 1, 0], np.uint16), sell_imbalances_sum=np.zeros(3, np.uint16),
               cot_price_levels=np.array([10, 12, 13], np.int32), imb_max_run_signed=np.array([1, 1, 0], np.int16),
               vp_skew=np.zeros(3), vp_gini=np.array([0.5, 0.6, 0.0]))
    return off, flat, bar


def test_footprint_data_container():
    from finmlkit_amd.bar.data_model import FootprintData
    off, flat, bar = _fake_csr()
    ts = np.array([60, 120, 180], dtype=np.int64) * 1_000_000_000
    fp = FootprintData.from_csr(ts, 0.5, off, flat, bar)
    assert len(fp) == 3 and fp.is_valid()
    assert [len(x) for x in fp.price_levels] == [2, 3, 1]
    assert fp.buy_imbalances[1].dtype == np.bool_ and fp.buy_volumes[0].dtype == np.float32
    assert fp.price_levels[1].base is not None            # views into the CSR buffer, not copies
    sub = fp[1:]
    assert len(sub) == 2 and sub.cot_price_levels.tolist() == [12, 13]
    one = fp[0:1]
    assert len(one) == 1
    with pytest.raises(TypeError):
        fp["x"]
    fp.cast_to_numba_list()
    assert isinstance(fp.price_levels, list)
    fp.cast_to_numpy()
    assert fp.price_levels.dtype == object
    df = fp.get_df()
    assert list(df.columns) == ["price_level", "sell_ticks", "buy_ticks", "sell_volume", "buy_volume",
                                "sell_imbalance", "buy_imbalance"]
    assert df.index.names == ["bar_idx", "bar_datetime_idx"] and len(df) == 6
    first_bar = df.xs(0, level="bar_idx")["price_level"].tolist()
    assert first_bar == [5.5, 5.0]                         # descending price inside a bar, scaled by the tick
    assert fp.memory_usage() > 0 and "Number of Bars: 3" in repr(fp)


def test_transform_naming():
    from finmlkit_amd.feature.transforms import EWMST, Compose, ReturnT
    r = ReturnT(pd.Timedelta(seconds=5), is_log=True, input_col="price")
    assert r.output_name == "price_ret5.0s" and ReturnT().produces == ["ret1"]
    e = EWMST(pd.Timedelta(minutes=1))
    assert e.output_name == "y_ewms60.0s"
    assert Compose(r, e).output_name == "price_ret5.0s_ewms60.0s"
    with pytest.raises(TypeError):
        r(np.zeros(3))
    with pytest.raises(ValueError, match="not found"):
        r(pd.DataFrame({"close": [1.0]}, index=pd.to_datetime([0])))


def test_plan_edges():
    from finmlkit_amd.dist import plan_edges
    e0, d, ne = 0, 10, 11                                   # edges 0,10,...,100
    plans = plan_edges([1, 35, 70], ne, e0, d)
    assert [(p.lo, p.hi) for p in plans] == [(0, 3), (3, 6), (6, 10)]
    assert sum(p.n_bars for p in plans) == ne - 1
    # a first timestamp exactly on an edge: that edge closes on the rank that holds the tick
    plans = plan_edges([1, 40], ne, e0, d)
    assert [(p.lo, p.hi) for p in plans] == [(0, 3), (3, 10)]
    with pytest.raises(ValueError, match="complete bar"):
        plan_edges([1, 12, 14], ne, e0, d)                  # middle shard holds no edge
    assert plan_edges([5], ne, e0, d)[0].n_bars == 10


def test_close_indices_out_of_range_are_refused_before_any_device_work():
    """bar_close_indices at or past the end of the arrays (IndexError in the reference's Python mode, an out-of-bounds read
    under Numba) must not reach the kernels; trade-size features are exempt (slice semantics).  Runs without a GPU."""
    from finmlkit_amd.bar import base
    px = np.arange(6, dtype=np.float64) + 100.0
    am = np.ones(6)
    sd = np.ones(6, np.int8)
    base._check_close_indices(np.array([0]), 0)          # one element = zero bars: nothing is indexed (the reference's
    base._check_close_indices(np.array([7]), 3)          # test_comp_bar_footprints_empty_bar passes exactly that)
    for ci in (np.array([-1, 2, 6]), np.array([0, 9]), np.array([-2, 5])):
        with pytest.raises(IndexError, match="out of bounds"):
            base.comp_bar_ohlcv(px, am, ci)
        with pytest.raises(IndexError, match="out of bounds"):
            base.comp_bar_directional_features(px, am, ci, sd)
        with pytest.raises(IndexError, match="out of bounds"):
            base.comp_bar_footprints(px, am, ci, sd, 0.5, px[:len(ci) - 1], px[:len(ci) - 1], 3.0)


def test_device_logarithm_and_exp_are_the_hosts(tmp_path):
    """(exp: csrc/fmk_exp.h restates glibc's exp the same way, for ewmst's alpha = 1 - exp(-dt / half_life), volatility.py:178-179; the
    same program checks it on 2e6 of ewmst's own arguments, a sweep of [-0.75, 0.75), random bit patterns, results from the subnormals to
    the overflow threshold and the special values.)  The logarithm of tick returns on the device (csrc/fmk_log.h) is an operation-by-operation restatement of glibc's log in libm's FMA
    build -- BOTH branches: the table-free one around 1 and the 128-entry table (constants extracted from the host's libm by
    tools/extract_glibc_tables.py); the header is plain C as well, and tools/logratio_check.c runs THE SAME SOURCE against this
    host's log() -- the function the oracle calls -- on 2e6 price quotients, a sweep of the table-free interval, quotients of prices up
    to a factor 2^40 apart, random bit patterns over the whole double range (subnormals, negatives, NaNs), the special values and the
    neighbours of every power of two: no difference allowed.  (A host without FMA3 runs libm's generic build, which differs from the FMA
    build on ~6 arguments in 1e5: the contract is the FMA build.)"""
    import os
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    if "fma" not in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        pytest.skip("host without FMA3: its libm runs the generic build of log")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "logratio_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", os.path.join(root, "tools", "logratio_check.c"), "-o", exe, "-lm"])
    r = subprocess.run([exe, "2000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "whole double range: 0 of" in r.stdout
    assert "differ from the host's exp" in r.stdout and "exp, whole double range: 0 of" in r.stdout


def test_the_log_and_exp_tables_in_the_tree_are_the_hosts(tmp_path):
    """csrc/fmk_logtab.h and csrc/fmk_exptab.h are what tools/extract_glibc_tables.py reads from THIS host's libm.so.6 (the constants of glibc's log have not
    changed since 2.28; a host whose libm carries other constants would make the device differ from the oracle there)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if not os.path.exists(libm):
        pytest.skip("no libm.so.6 at the usual place")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "extract_glibc_tables.py"), libm, str(tmp_path)])
    strip = lambda t: [l for l in t.splitlines() if not l.startswith("//")]
    for name in ("fmk_logtab.h", "fmk_exptab.h"):
        assert strip(open(str(tmp_path / name)).read()) == strip(open(os.path.join(root, "finmlkit_amd", "csrc", name)).read()), name
