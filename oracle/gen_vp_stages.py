#!/usr/bin/env python3
"""tests/golden/volume_profile_stages.npz: the three helper functions of the rolling volume profile run by the REFERENCE itself
(finmlkit/feature/core/volume.py: aggregate_footprint :134-203, bucket_price_levels :207-275, comp_poc_hva_lva :278-365), in the
pure-Python mode its CI pins (see oracle/gen_golden.py).  Inputs: the footprints of the synthetic stream with sizes that are
multiples of 2^-4 (every float32 sum on the way is exact, so typed and pure-Python semantics coincide) -- regenerated in the
tests from the seed; stored: the windows / parameters asked for and the reference's outputs.  No reference source is copied.

    python oracle/gen_vp_stages.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)
os.environ["NUMBA_DISABLE_JIT"] = "1"

import numpy as np  # noqa: E402
from finmlkit.feature.core import volume as rvolume  # noqa: E402
from numba.typed import List as NList  # noqa: E402

from oracle import oracle as orc  # noqa: E402


def main():
    d = {}
    n = 60_000
    ts, px, _, sd = orc.synth(42, 0, n)
    am = (np.random.default_rng(5).integers(1, 65, n) * 2.0 ** -4).astype(np.float32)
    clock, ci = orc._time_bar_indexer(ts, 60.0)
    o = orc.comp_bar_ohlcv(px, am, ci, want_median=False)
    off, flat, _ = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, o[2], o[1], 3.0)
    nb = len(ci) - 1
    split = lambda a: NList([a[off[i]:off[i + 1]] for i in range(nb)])
    bar_ts = clock[1:]
    pls, bvs, svs = split(flat["price_levels"]), split(flat["buy_volumes"]), split(flat["sell_volumes"])
    # windows: a usual half hour, one bar, the one-bar fallback (start == end between two bars), the whole stream, the first bars
    wins = [(bar_ts[20] - 1_800_000_000_000, bar_ts[20]), (bar_ts[7], bar_ts[7]), (bar_ts[11] + 5, bar_ts[11] + 6),
            (bar_ts[0], bar_ts[-1]), (bar_ts[0] - 10**12, bar_ts[2]), (bar_ts[30] - 299_999_999_999, bar_ts[30])]
    d["windows"] = np.array(wins, dtype=np.int64)
    for w, (s, e) in enumerate(wins):
        lv, ab, as_ = rvolume.aggregate_footprint(bar_ts, o[1], o[2], pls, bvs, svs, int(s), int(e), 0.01)
        d[f"agg{w}__levels"], d[f"agg{w}__buy"], d[f"agg{w}__sell"] = lv, ab, as_
        tot = ab + as_
        for nbins in (27, 5, 200, 3):
            if len(lv) < 2:
                continue
            bl, bv = rvolume.bucket_price_levels(lv, tot, nbins)
            d[f"bkt{w}_{nbins}__levels"], d[f"bkt{w}_{nbins}__volumes"] = bl, bv
            for va in (68.34, 95.0):
                d[f"poc{w}_{nbins}_{va}"] = np.array(rvolume.comp_poc_hva_lva(bl, bv, va), dtype=np.int64)
        for va in (68.34, 30.0):
            d[f"poc{w}_raw_{va}"] = np.array(rvolume.comp_poc_hva_lva(lv, tot, va), dtype=np.int64)
    # comp_poc_hva_lva on INEXACT volumes: lognormal float32 profiles, regenerated in the tests from the seed -- the
    # reference in the same pinned pure-Python mode (NumPy's pairwise float32 np.sum; its np.float32 scalars).  Stored: the outputs.
    rng = np.random.default_rng(20260930)
    res = []
    for _ in range(400):
        m = int(rng.integers(1, 300))
        v = rng.lognormal(0.0, 1.5, m).astype(np.float32)
        va = float(rng.choice([68.34, 50.0, 95.0]))
        res.append(rvolume.comp_poc_hva_lva(np.arange(m, dtype=np.int32) * 3 + 1000, v, va))
    d["poc_lognormal"] = np.array(res, dtype=np.int64)
    d["synth"] = np.array([42, 0, n, orc.DENSE_GAP_MOD], dtype=np.int64)
    d["amount"] = am
    path = os.path.join(ROOT, "tests", "golden", "volume_profile_stages.npz")
    np.savez_compressed(path, **d)
    print(f"volume_profile_stages: {len(d)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
