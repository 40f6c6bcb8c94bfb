"""GPU parity of the rolling volume profile ("next" rank 2): VolumePro / volume_profile_rolling vs reference-generated
goldens and the CPU oracle."""
import numpy as np
import pandas as pd
import pytest

from tests import _golden as G
from tests.test_oracle_golden import _vp_inputs

pytestmark = pytest.mark.gpu

KEYS = ("poc", "hva", "lva", "pct")


def test_volume_profile_golden(orc):
    from finmlkit_amd.feature.core.volume import volume_profile_rolling, volume_profile_rolling_csr
    d = G.load("volume_profile")
    for name in ("m1_w30", "m1_w5_nobins", "s10_w120_b5", "m1_w30_b200"):
        bts, hi, lo, off, flat, window, nbins, va = _vp_inputs(orc, d, name)
        got = volume_profile_rolling_csr(bts, hi, lo, off, flat["price_levels"], flat["buy_volumes"],
                                         flat["sell_volumes"], window, nbins, 0.01, va)
        for g, k in zip(got, KEYS):
            assert g.dtype == d[f"{name}__{k}"].dtype
            np.testing.assert_array_equal(g, d[f"{name}__{k}"], err_msg=f"{name}:{k}")
        split = lambda a: [a[off[i]:off[i + 1]] for i in range(len(off) - 1)]          # the reference's ragged form
        got2 = volume_profile_rolling(bts, hi, lo, split(flat["price_levels"]), split(flat["buy_volumes"]),
                                      split(flat["sell_volumes"]), window, nbins, 0.01, va)
        for a, b in zip(got, got2):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n,interval,window,nbins,inexact,tick", [
    (600_000, 60.0, 1800.0, 27, False, 0.01), (600_000, 60.0, 1800.0, 27, True, 0.01),
    (400_000, 5.0, 60.0, None, True, 0.01),
    (400_000, 60.0, 14_400.0, 11, True, 0.01),          # > 1024 levels per window: one wave per workgroup
    (300_000, 1.0, 30.0, 3, False, 0.01),
    (500_000, 600.0, 7200.0, 27, True, 0.0001)])        # > 8192 levels per window: histogram in global scratch
def test_volume_profile_vs_oracle(orc, n, interval, window, nbins, inexact, tick):
    from finmlkit_amd.feature.core.volume import volume_profile_rolling_csr
    ts, px, am, sd = orc.synth(23, 0, n)
    if inexact:
        am = np.random.default_rng(1).lognormal(-1, 1.0, n).astype(np.float32)    # float32 sums round: order matters
    clock, ci = orc._time_bar_indexer(ts, interval)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    off, flat, _ = orc.comp_bar_footprints_csr(px, am, ci, sd, tick, o[2], o[1], 3.0)
    args = (clock[1:], o[1], o[2], off, flat["price_levels"], flat["buy_volumes"], flat["sell_volumes"], window, nbins,
            tick, 68.34)
    want = orc.volume_profile_rolling(*args)
    got = volume_profile_rolling_csr(*args)
    for g, w, k in zip(got, want, KEYS):
        np.testing.assert_array_equal(g, w, err_msg=k)
    assert (want[0] != 0).sum() > 10


def test_volumepro_on_kit_output_and_errors(orc):
    from finmlkit_amd.bar.data_model import TradesData
    from finmlkit_amd.bar.kit import TimeBarKit
    from finmlkit_amd.feature.core.volume import VolumePro, volume_profile_rolling_csr
    n = 300_000
    ts, px, am, sd = orc.synth(3, 0, n)
    kit = TimeBarKit(TradesData(ts, px, am, np.arange(n), side=sd), pd.Timedelta(seconds=60))
    bars = kit.build_ohlcv()
    fp = kit.build_footprints(price_tick_size=0.01)
    vp = VolumePro(pd.Timedelta(minutes=20), n_bins=15)
    poc, hva, lva, pct = vp.compute(bars, fp)
    want = orc.volume_profile_rolling(fp.bar_timestamps, bars.high.values, bars.low.values, fp.level_offsets,
                                      fp.flat["price_levels"], fp.flat["buy_volumes"], fp.flat["sell_volumes"],
                                      1200.0, 15, 0.01, 68.34)
    first = int(np.flatnonzero(want[0])[0])
    assert np.isnan(poc[:first]).all() and not np.isnan(poc[first:]).any()
    np.testing.assert_array_equal(poc[first:], want[0][first:] * 0.01)
    np.testing.assert_array_equal(hva[first:], want[1][first:] * 0.01)
    np.testing.assert_array_equal(lva[first:], want[2][first:] * 0.01)
    np.testing.assert_array_equal(pct, want[3])
    assert np.all(lva[first:] <= poc[first:]) and np.all(poc[first:] <= hva[first:])
    # a slice loses the CSR view and goes through the ragged-list path: same numbers
    t0, t1 = pd.to_datetime(fp.bar_timestamps[40], unit="ns"), pd.to_datetime(fp.bar_timestamps[120], unit="ns")
    bts, poc_r, hva_r, lva_r, pct_r = vp.compute_range(bars, fp, t0, t1)
    assert bts[-1] == fp.bar_timestamps[120] and len(poc_r) == len(bts)
    with pytest.raises(ZeroDivisionError):
        volume_profile_rolling_csr(fp.bar_timestamps, bars.high.values, bars.low.values, fp.level_offsets,
                                   fp.flat["price_levels"], fp.flat["buy_volumes"], fp.flat["sell_volumes"], 1200.0, 0,
                                   0.01)


def test_pct_above_poc_nan_total_propagates(orc):
    """volume.py:378 guards with `total_volume <= 0`, which a NaN total does not trip: NaN comes out (not 0.0); an all-zero
    profile and a profile with nothing above the POC give 0.0."""
    from finmlkit_amd.feature.core import volume
    pl = np.array([1, 2, 3, 4], np.int32)
    v = np.array([1.0, np.nan, 2.0, 1.0], np.float32)
    assert np.isnan(orc.calc_volume_percentage_above_poc(pl, v, 2))
    assert np.isnan(volume.calc_volume_percentage_above_poc(pl, v, 2))
    z = np.zeros(4, np.float32)
    assert volume.calc_volume_percentage_above_poc(pl, z, 2) == 0.0 == orc.calc_volume_percentage_above_poc(pl, z, 2)
    w = np.array([1.0, 2.0, 0.0, 0.0], np.float32)
    assert volume.calc_volume_percentage_above_poc(pl, w, 2) == 0.0 == orc.calc_volume_percentage_above_poc(pl, w, 2)


def test_volume_profile_stages_golden(orc):
    """aggregate_footprint / bucket_price_levels / comp_poc_hva_lva as callable API (VERDICT r4 missing #3): vectors made by the
    reference itself (oracle/gen_vp_stages.py), bit for bit; the ragged-list and the CSR form of the first agree."""
    from finmlkit_amd.feature.core.volume import aggregate_footprint, bucket_price_levels, comp_poc_hva_lva
    d = G.load("volume_profile_stages")
    ts, px, _, sd = G.synth_from(orc, d["synth"])
    am = d["amount"]
    clock, ci = orc._time_bar_indexer(ts, 60.0)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    off, flat, _ = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    bar_ts = clock[1:]
    split = lambda a: [a[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    ragged = (split(flat["price_levels"]), split(flat["buy_volumes"]), split(flat["sell_volumes"]))
    n_agg = n_bkt = n_poc = 0
    for w, (s, e) in enumerate(d["windows"]):
        lv, ab, as_ = aggregate_footprint(bar_ts, o[1], o[2], flat["price_levels"], flat["buy_volumes"], flat["sell_volumes"],
                                          int(s), int(e), 0.01, level_offsets=off)
        for g, k in zip((lv, ab, as_), ("levels", "buy", "sell")):
            assert g.dtype == d[f"agg{w}__{k}"].dtype
            np.testing.assert_array_equal(g, d[f"agg{w}__{k}"], err_msg=f"agg{w}:{k}")
        if w < 3:
            for a, b in zip((lv, ab, as_), aggregate_footprint(bar_ts, o[1], o[2], *ragged, int(s), int(e), 0.01)):
                np.testing.assert_array_equal(a, b)
        n_agg += 1
        tot = ab + as_
        for key in [k for k in d if k.startswith(f"bkt{w}_") and k.endswith("__levels")]:
            nbins = int(key.split("_")[1])
            bl, bv = bucket_price_levels(lv, tot, nbins)
            assert bl.dtype == np.int32 and bv.dtype == np.float32
            np.testing.assert_array_equal(bl, d[key], err_msg=key)
            np.testing.assert_array_equal(bv, d[f"bkt{w}_{nbins}__volumes"], err_msg=key)
            n_bkt += 1
            for va in (68.34, 95.0):
                assert comp_poc_hva_lva(bl, bv, va) == tuple(int(x) for x in d[f"poc{w}_{nbins}_{va}"]), (key, va)
                n_poc += 1
        for va in (68.34, 30.0):
            assert comp_poc_hva_lva(lv, tot, va) == tuple(int(x) for x in d[f"poc{w}_raw_{va}"]), (w, va)
            n_poc += 1
    assert n_agg == 6 and n_bkt >= 16 and n_poc >= 40
    # inexact volumes (lognormal float32 profiles from the seed of oracle/gen_vp_stages.py): the reference's own answers in its pinned
    # pure-Python mode.  The library's contract (include/fmk.h): NumPy's pairwise float32 np.sum for the total, float64 scalars for
    # the walk -- on these 400 profiles every reading of the reference's scalars gives the same three levels.
    rng = np.random.default_rng(20260930)
    for want in d["poc_lognormal"]:
        m = int(rng.integers(1, 300))
        v = rng.lognormal(0.0, 1.5, m).astype(np.float32)
        va = float(rng.choice([68.34, 50.0, 95.0]))
        assert comp_poc_hva_lva(np.arange(m, dtype=np.int32) * 3 + 1000, v, va) == tuple(int(x) for x in want), (m, va)
    with pytest.raises(ZeroDivisionError):
        bucket_price_levels(np.arange(5, dtype=np.int32), np.ones(5, dtype=np.float32), 0)
    with pytest.raises(ValueError):
        bucket_price_levels(np.array([7], dtype=np.int32), np.ones(1, dtype=np.float32), 27)      # one level: the reference's broadcast error
