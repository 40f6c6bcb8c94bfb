#!/usr/bin/env python3
"""The tick-level chain through the REFERENCE's own loops: comp_lagged_returns (feature/core/utils.py:12-64) -> ewmst
(feature/core/volatility.py:139-219) -> _cusum_bar_indexer (bar/logic.py:152-221) on the first N ticks of the seed-42 stream of
SURVEY.md 8(d).

Runs only in the build container (needs /root/reference; pure-Python mode as the reference's CI pins it, see oracle/gen_golden.py).
Stored (data only): every 29th value of the returns and of sigma, their NaN counts, all CUSUM close indices.  The oracle is
checked against it on the CPU (tests/test_oracle_golden.py), the HIP kernels on the GPU (tests/test_gpu_ticklevel.py).

Split out of oracle/gen_cfg1.py in round 3 (there it ran at 10^6 ticks and took ~10 of that script's 12 minutes; the pure-Python
loops do ~1 700 ticks per second): N = 300 000 keeps the whole regeneration at ~3 minutes, the denser sampling (29 instead of 97)
keeps about as many compared values.

    python oracle/gen_ticklevel_chain.py            # ~3 min; rewrites tests/golden/ticklevel_chain_reference.npz
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)
os.environ["NUMBA_DISABLE_JIT"] = "1"

import numpy as np  # noqa: E402

import finmlkit.bar.logic as LG  # noqa: E402
from finmlkit.feature.core import utils as FU  # noqa: E402
from finmlkit.feature.core import volatility as FV  # noqa: E402

from oracle import oracle as orc  # noqa: E402

N = 300_000
STEP = 29


def main():
    orc.build()
    t0 = time.time()
    ts, px, am, sd = orc.synth(42, 0, N)
    r = FU.comp_lagged_returns(ts, px, 5.0, True)
    sg = FV.ewmst(ts, r, 60.0)
    ci = np.array(LG._cusum_bar_indexer(ts, px, sg.copy(), 1e-5, 2.0), dtype=np.int64)
    d = {"seed": np.int64(42), "n": np.int64(N), "step": np.int64(STEP), "return_window_sec": np.float64(5.0),
         "half_life_sec": np.float64(60.0), "sigma_floor": np.float64(1e-5), "lambda_mult": np.float64(2.0),
         "returns_sampled": r[::STEP].copy(), "sigma_sampled": sg[::STEP].copy(),
         "returns_nan": np.int64(np.isnan(r).sum()), "sigma_nan": np.int64(np.isnan(sg).sum()),
         "cusum_close_indices": ci}
    print(f"lagged returns -> ewmst -> CUSUM on {N} ticks: {len(ci) - 1} bars, in {time.time() - t0:.0f} s")
    path = os.path.join(ROOT, "tests", "golden", "ticklevel_chain_reference.npz")
    np.savez_compressed(path, **d)
    print(f"{path}: {len(d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
