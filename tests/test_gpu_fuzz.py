"""A short fixed-seed campaign of the randomized differential test (tools/fuzz_parity.py): the HIP path against the oracle
on random sizes biased towards the kernels' internal boundaries, random bar structures, dtypes, thresholds, windows and NaN
placements, every function under the comparison policy of tests/_refcalls.py.  21 000 cases over eight seeds were run while
round 1 was built (one failure: an uncertified dollar-bar decision, now redone by the exact loop)."""
import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_parity_fixed_seed(orc):
    from tools.fuzz_parity import campaign
    fails = campaign(600, 20260928, orc, verbose=False)
    assert not fails, "\n".join(fails[:10])


def test_fuzz_long_bars_fixed_seed(orc):
    """tools/fuzz_longbars.py: the four bar reducers on bars around every threshold of the workgroup-per-bar schedules (8 192 ..
    65 536 ticks -+ 1) mixed with short, empty and very long ones; dyadic, full-mantissa float32 and float64 amounts, a NaN or a
    negative size now and then, coarse to fine price grids (240 cases over two seeds were run when the schedules were built)."""
    from tools.fuzz_longbars import campaign
    fails = campaign(16, 20260929, orc, verbose=False)
    assert not fails, "\n".join(fails[:5])


@pytest.mark.gpu
def test_fuzz_mid_length_bars_fixed_seed(orc):
    """... and on the lengths between the two ends (128 .. 40 000 ticks: every register class of the medians and footprints, the
    one-read trade-size kernels with 1, 2, 4, 8 and 16 waves per bar -+ 1 tick around each of their edges)."""
    from tools.fuzz_longbars import campaign
    fails = campaign(24, 20260930, orc, verbose=False, mid=True)
    assert not fails, "\n".join(fails[:5])


def test_fuzz_short_bar_streams_fixed_seed(orc):
    """... and on STREAMS of 17 000 .. 40 000 short bars (geometric lengths around 12 .. 200 ticks with the schedules' edges mixed in:
    64 / 65, 128 / 129, 248 / 249 .. 255 / 256 / 257): what selects the lane-per-bar and sixteen-lanes-per-bar schedules and their
    hand-over lists, for all four reducers (460 cases over two seeds were run when the mode was added)."""
    from tools.fuzz_longbars import campaign
    fails = campaign(8, 20261001, orc, verbose=False, mid="short")
    assert not fails, "\n".join(fails[:5])


def test_fuzz_sharded_fixed_seed():
    """tools/fuzz_sharded.py: random world sizes (2..8 virtual ranks), ticks per rank, stream density and bar interval; the
    concatenated per-rank outputs of the sharded time-bar step equal the un-sharded run bit for bit (60 configurations of
    seed 1 were run while round 1 was built)."""
    from tools.fuzz_sharded import sweep
    fails, ran = sweep(10, 20260928, verbose=False)
    assert ran >= 8 and not fails, "\n".join(fails[:5])


@pytest.mark.parametrize("kind,seed", [("volume", 1), ("volume", 2), ("dollar", 1)])
def test_fuzz_threshold_indexers_fixed_seed(orc, kind, seed):
    """tools/fuzz_volume.py: streams up to 1e6 ticks of lognormal / decimal / integer / quarter lots, zeros, whales, NaN and
    negative amounts or prices; thresholds as multiples of the mean (bar lengths 0.5 .. 3e5 ticks: every tier), round
    numbers, some tick's running sum to the last bit, the total.  Exact mode equals the oracle with n_uncertified == 0; the
    fast mode may only differ when it reports a decision.  (Seed 2 holds the NaN case that exposed the poisoned exclusive
    prefix of the chunked serial walk.)"""
    from tools.fuzz_volume import run
    bad, reported, fast_diff = run(seed, 120, 1_000_000, kind, verbose=False)
    assert bad == 0
    print(f"{kind} seed {seed}: fast mode reported decisions in {reported} of 120 cases, differed from the reference in {fast_diff}")
