"""CPU, world_size = 2, 3 and 8 as real processes: the sharding plan of finmlkit_amd/dist.py and the C entry points of
csrc/fmk_comm.hip (`fmk_comm_create / _allgather / _barrier / _halo_exchange_dev`) over the HOST-STAGED transport with
ctx = NULL (the pointers are NumPy buffers) reproduce the single-process result exactly.  The per-bar arithmetic is the
CPU oracle's here; on the GPU box tests/test_gpu_dist.py runs the same plan + entry points with the HIP kernels (two
processes on one device), and bench.py the RCCL transport.  No PyTorch."""
import multiprocessing as mp
import os

import numpy as np
import pytest

INTERVAL = 60.0
N_TOTAL = 90_000


def _worker(rank, world, path, gap_mod, out_dir, ring_bytes):
    from finmlkit_amd.dist import Comm, plan_edges
    from oracle import oracle as orc

    comm = Comm(None, rank, world, path, "host", ring_bytes=ring_bytes, timeout_s=60.0)
    n = N_TOTAL // world
    cols = list(orc.synth(42, rank * n, n, gap_mod))                      # my shard of ONE global stream
    ts = cols[0]

    allfl = comm.all_gather_i64([int(ts[0]), int(ts[-1])])
    ne, e0, d = orc.time_bar_clock(allfl[0][0], allfl[-1][1], INTERVAL)   # global clock
    plans = plan_edges([a[0] for a in allfl], ne, e0, d)
    my = plans[rank]
    c_last = int(np.searchsorted(ts, e0 + my.hi * d, side="right")) - 1   # local close of my last edge
    send_h = n - c_last if rank + 1 < world else 0
    allh = comm.all_gather_i64([send_h])
    recv_h = allh[rank - 1][0] if rank > 0 else 0
    ext = [np.empty(recv_h + n, c.dtype) for c in cols]                   # [halo | shard]
    for e, c in zip(ext, cols):
        e[recv_h:] = c
    send = [(c[n - send_h:].ctypes.data, send_h * c.itemsize) for c in cols]
    recv = [(e.ctypes.data, recv_h * e.itemsize) for e in ext]
    for _ in range(2):                                                    # a second round reuses the rings
        comm.exchange(send, recv)
        comm.wait()
    comm.barrier()
    ets, epx, eam, esd = ext
    edges = e0 + np.arange(my.lo, my.hi + 1, dtype=np.int64) * d
    ci = np.searchsorted(ets, edges, side="right").astype(np.int64) - 1
    if rank > 0:
        assert ci[0] == 0                                                 # the halo's first tick is the open edge
    res = {"edges": edges[1:]}
    for k, v in zip(["open", "high", "low", "close", "volume", "vwap", "trades", "median"],
                    orc.comp_bar_ohlcv(epx, eam, ci)):
        res["o_" + k] = v
    if gap_mod == orc.DENSE_GAP_MOD:                                       # no empty bars -> defined everywhere
        for i, v in enumerate(orc.comp_bar_directional_features(epx, eam, ci, esd)):
            res[f"d_{i}"] = v
        off, flat, bar = orc.comp_bar_footprints_csr(epx, eam, ci, esd, 0.01, res["o_low"], res["o_high"], 3.0)
        res["f_nlev"] = np.diff(off)
        for k, v in {**flat, **bar}.items():
            res["f_" + k] = v
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    t = comm.all_gather_f64([float(rank) + 0.5])
    assert [x[0] for x in t] == [r + 0.5 for r in range(world)]
    comm.close()


def _spawn(target, world, args):
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=target, args=(r, world) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("worker hung")
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"


# ring of 4 KiB: the ~15-40 KB halo wraps the ring many times (chunked progress); 1 MiB: one piece
# (world 8: the size of the node the driver scales to -- seven neighbour exchanges, eight-row gathers)
@pytest.mark.parametrize("world,sparse,ring", [(2, False, 4096), (3, False, 1 << 20), (2, True, 4096), (8, False, 4096)])
def test_sharded_time_bars_match_single_process(tmp_path, orc, world, sparse, ring):
    gap = orc.SPARSE_GAP_MOD if sparse else orc.DENSE_GAP_MOD
    _spawn(_worker, world, (str(tmp_path / "rdv"), gap, str(tmp_path), ring))
    assert not (tmp_path / "rdv").exists()                               # rank 0 unlinked the rendezvous file
    parts = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(world)]
    n = (N_TOTAL // world) * world
    ts, px, am, sd = orc.synth(42, 0, n, gap)
    clock, ci = orc._time_bar_indexer(ts, INTERVAL)
    cat = lambda k: np.concatenate([p[k] for p in parts])
    np.testing.assert_array_equal(cat("edges"), clock[1:])
    want = orc.comp_bar_ohlcv(px, am, ci)
    for k, w in zip(["open", "high", "low", "close", "volume", "vwap", "trades", "median"], want):
        np.testing.assert_array_equal(cat("o_" + k), w, err_msg=k)        # identical, not merely close
    if not sparse:
        for i, w in enumerate(orc.comp_bar_directional_features(px, am, ci, sd)):
            got = cat(f"d_{i}")
            if i in (6, 7):    # spread of the very first bar uses the wrap-around tick prices[-1] (reference quirk):
                got, w = got[1:], w[1:]   # undefined across shards, identical everywhere else
            np.testing.assert_array_equal(got, w, err_msg=f"dir {i}")
        off, flat, bar = orc.comp_bar_footprints_csr(px, am, ci, sd, 0.01, want[2], want[1], 3.0)
        np.testing.assert_array_equal(cat("f_nlev"), np.diff(off))
        for k, w in {**flat, **bar}.items():
            np.testing.assert_array_equal(cat("f_" + k), w, err_msg=k)


def test_self_loop_and_errors(tmp_path):
    """world = 1 with the self-loop flag: the rank is its own neighbour, the message is longer than the ring; and the
    failure modes are error codes, not hangs."""
    from finmlkit_amd import _ffi
    from finmlkit_amd.dist import Comm
    comm = Comm(None, 0, 1, str(tmp_path / "loop"), "host", self_loop=True, ring_bytes=4096)
    a = np.arange(50_000, dtype=np.int64)
    b = np.zeros_like(a)
    c8 = (np.arange(777) % 251).astype(np.int8)
    d8 = np.zeros_like(c8)
    comm.exchange([(a.ctypes.data, a.nbytes), (0, 0), (c8.ctypes.data, c8.nbytes)],
                  [(b.ctypes.data, b.nbytes), (0, 0), (d8.ctypes.data, d8.nbytes)])
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(c8, d8)
    assert comm.all_gather_i64([7, 8]) == [[7, 8]]
    with pytest.raises(ValueError):
        comm.all_gather_i64(list(range(1000)))                           # > 4096 bytes per rank
    comm.close()
    with pytest.raises(_ffi.FmkError, match="did not appear"):           # rank 1 of 2 and nobody creates the segment
        Comm(None, 1, 2, str(tmp_path / "nobody"), "host", timeout_s=0.5)
    with pytest.raises(ValueError):
        Comm(None, 3, 2, str(tmp_path / "bad"), "host")
    with pytest.raises(ValueError, match="needs a context"):
        Comm(None, 0, 1, str(tmp_path / "bad2"), "rccl")


@pytest.mark.parametrize("n,thr", [(1000, 7), (1000, 1), (1000, 0), (1, 5), (64, 64), (65, 64), (10_000, 2), (999, 1000), (5, 2)])
def test_sharded_tick_bars_concatenate_to_the_reference(orc, n, thr):
    """tick bars shard without communication -- the closes of a shard are a closed form of its global tick
    range; concatenated they are `_tick_bar_indexer`'s (logic.py:54-84), whatever the cut points."""
    from finmlkit_amd.dist import sharded_tick_bar_index
    want = orc._tick_bar_indexer(np.arange(n, dtype=np.int64), thr)
    rng = np.random.default_rng(n * 31 + thr)
    for world in (1, 2, 3, 8):
        cuts = np.sort(rng.integers(0, n + 1, size=world - 1)) if world > 1 else np.zeros(0, np.int64)
        edges = np.concatenate([[0], cuts, [n]]).astype(np.int64)
        got = np.concatenate([sharded_tick_bar_index(edges[r], edges[r + 1] - edges[r], thr) for r in range(world)])
        np.testing.assert_array_equal(got, want, err_msg=f"world {world} cuts {cuts}")


def _stale_worker(rank, world, path, delay):
    import time
    from finmlkit_amd.dist import Comm
    time.sleep(delay)
    comm = Comm(None, rank, world, path, "host", ring_bytes=4096, timeout_s=30.0)
    got = comm.all_gather_i64([100 + rank])
    assert [g[0] for g in got] == [100 + r for r in range(world)], got
    comm.profile_enable(True)
    a = np.arange(3000, dtype=np.int64) + rank
    b = np.zeros_like(a)
    comm.exchange([(a.ctypes.data, a.nbytes)], [(b.ctypes.data, b.nbytes)])
    if rank > 0:
        np.testing.assert_array_equal(b, np.arange(3000, dtype=np.int64) + rank - 1)
    ms = comm.profile_read()
    assert len(ms) == 1 and ms[0] >= 0.0
    comm.barrier()
    comm.close()


def test_leftover_rendezvous_of_a_dead_run_is_replaced(tmp_path):
    """A run that died before its ranks had all attached leaves its segment behind -- valid magic, same world and ring size, old
    gather generations.  Rank 1 of the NEXT run finds it first (rank 0 starts later): it must not pass its first all-gather on
    the dead run's data; rank 0 replaces the file (new inode, rename into place) and rank 1 re-attaches to the new one."""
    from finmlkit_amd.dist import Comm
    path = str(tmp_path / "rdv")
    # the dead run: rank 0 of world 2 creates the segment, publishes a gather generation, and is never joined
    ctx = mp.get_context("spawn")
    dead = ctx.Process(target=_dead_rank0, args=(path,))
    dead.start()
    dead.join(60)
    assert dead.exitcode == 0 and os.path.exists(path)
    _spawn_delayed(_stale_worker, 2, path, delays=[1.0, 0.0])             # rank 1 first, rank 0 a second later
    assert not os.path.exists(path)


def _dead_rank0(path):
    from finmlkit_amd import _ffi
    from finmlkit_amd.dist import Comm
    try:
        Comm(None, 0, 2, path, "host", ring_bytes=4096, timeout_s=0.3)   # nobody joins: set-up times out, the file stays
    except _ffi.FmkError:
        pass
    os._exit(0)


def _spawn_delayed(target, world, path, delays):
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=target, args=(r, world, path, delays[r])) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("worker hung")
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"


def test_rendezvous_does_not_follow_symlinks(tmp_path):
    from finmlkit_amd import _ffi
    from finmlkit_amd.dist import Comm
    target = tmp_path / "victim"
    target.write_bytes(b"\0" * (1 << 20))
    link = tmp_path / "rdv"
    os.symlink(target, link)
    with pytest.raises(_ffi.FmkError):                                    # rank 1 opens with O_NOFOLLOW
        Comm(None, 1, 2, str(link), "host", ring_bytes=4096, timeout_s=0.5)
    assert target.read_bytes() == b"\0" * (1 << 20)
