#!/usr/bin/env python3
"""One case of a tools/fuzz_volume.py campaign again (the generator's draws do not depend on the device): the indexer under the
tier knobs, which path answered, and the inputs as .npz for a look on the CPU.   usage: fuzz_case.py seed case [max_n] [volume|dollar]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import fuzz_volume as fv
from oracle import oracle as orc


def regenerate(seed, case, max_n, kind):
    rng = np.random.default_rng(seed)
    for c in range(case + 1):
        n = int(np.exp(rng.uniform(np.log(1), np.log(max_n))))
        if rng.random() < 0.15:
            n = int(rng.choice([1, 2, 511, 512, 513, 2047, 2048, 2049, 4096, 4097, 65536, 65537, 524288, 524289]))
            n = min(n, max_n)
        dist, a = fv.amounts(rng, n)
        if kind == "dollar":
            px = fv.prices(rng, n)
            thr = fv.threshold(rng, a.astype(np.float64) * px)
        else:
            px = np.ones(n)
            thr = fv.threshold(rng, a)
    return dist, a, px, thr


if __name__ == "__main__":
    seed, case = int(sys.argv[1]), int(sys.argv[2])
    max_n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 3_000_000
    kind = sys.argv[4] if len(sys.argv) > 4 else "volume"
    dist, a, px, thr = regenerate(seed, case, max_n, kind)
    want = orc._dollar_bar_indexer(px, a, thr) if kind == "dollar" else orc._volume_bar_indexer(a, thr)
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    np.savez_compressed(os.path.join(out, f"fuzz_{kind}_{seed}_{case}.npz"), a=a, px=px, thr=thr, want=want)
    print(f"{kind} seed {seed} case {case}: {dist} {a.dtype} n={len(a)} thr={thr!r} ({len(want) - 1} bars)")
    from finmlkit_amd import _ffi, engine
    ctx = _ffi.default_context()
    t = engine.DeviceTrades.from_numpy(np.arange(len(a), dtype=np.int64), px, a)
    index = t.dollar_bar_index if kind == "dollar" else t.volume_bar_index
    for fast in (False, True):
        ctx.set_fast_threshold(fast)
        got = index(thr).to_host()
        m = min(len(got), len(want))
        bad = np.flatnonzero(got[:m] != want[:m])
        print(f"  fast={fast}: {len(got)} closes, uncertified {t.last_uncertified}, first difference at "
              f"{int(bad[0]) if len(bad) else None}: {got[bad[0] - 1:bad[0] + 2] if len(bad) else ''} vs {want[bad[0] - 1:bad[0] + 2] if len(bad) else ''}")
    ctx.set_fast_threshold(False)
