#!/usr/bin/env python3
"""Replays ONE case of tools/fuzz_parity.py (seed, case, hi) that lands on _volume_bar_indexer and prints both close lists
with the running sums around the first difference.    python tools/volcase.py seed case [hi]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tools import fuzz_parity as F

seed, case = int(sys.argv[1]), int(sys.argv[2])
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
rng = np.random.default_rng([seed, case])
n = F.size(rng, hi)
ts, px, am, sd = F.tape(rng, n)
ci = F.bars(rng, n)
which = int(rng.integers(0, 18))
assert which == 1, which
thr = float(np.mean(am, dtype=np.float64)) * float(rng.choice([0.5, 3, 50, 700, 1500, 2500, 5000, 10**7]))
thr = F.knife_edge(rng, am.astype(np.float64), thr)
print(f"n={n} dtype={am.dtype} thr={thr!r} mean ticks/bar {thr / float(np.mean(am, dtype=np.float64)):g}")
from oracle import oracle as orc
from finmlkit_amd.bar import logic
want = orc._volume_bar_indexer(am, thr)
for k in range(3):
    got = logic._volume_bar_indexer(am, thr)
    same = len(got) == len(want) and np.array_equal(got, want)
    print(f"run {k}: {len(got)} closes vs {len(want)}: {'identical' if same else 'DIFFERENT'}")
    if not same:
        m = min(len(got), len(want))
        d = int(np.flatnonzero(got[:m] != want[:m])[0]) if m and (got[:m] != want[:m]).any() else m
        print("  first difference at bar", d, "got", got[max(0, d - 1):d + 3], "want", want[max(0, d - 1):d + 3])
        s = 0.0
        a0 = int(want[d - 1]) + 1 if d > 0 else 0
        for i in range(a0, int(max(got[d], want[d])) + 1):
            s += float(am[i])
            if i >= min(got[d], want[d]) - 1:
                print(f"    tick {i}: amount {float(am[i])!r} running {s!r} (threshold {thr!r}, diff {s - thr:.3e})")
