"""CPU, world_size = 2 and 3 over gloo: the sharding plan + neighbour halo exchange of
finmlkit_amd/dist.py reproduce the single-process result exactly (the per-bar arithmetic is done by
the CPU oracle here -- on the GPU box bench.py drives the same plan/exchange with the HIP kernels)."""
import os
import socket

import numpy as np
import pytest

INTERVAL = 60.0
N_TOTAL = 90_000
HALO = 4096


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, gap_mod, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from finmlkit_amd.dist import Comm, halo_lengths, plan_edges
    from oracle import oracle as orc

    comm = Comm()
    n = N_TOTAL // world
    cols = list(orc.synth(42, rank * n, n, gap_mod))                      # my shard of ONE global stream
    back = [np.zeros(HALO + n, c.dtype) for c in cols]                    # [headroom | shard]
    for b, c in zip(back, cols):
        b[HALO:] = c
    tens = [torch.from_numpy(b) for b in back]
    ts = cols[0]

    allfl = comm.all_gather_i64([int(ts[0]), int(ts[-1])])
    ne, e0, d = orc.time_bar_clock(allfl[0][0], allfl[-1][1], INTERVAL)   # global clock
    plans = plan_edges([a[0] for a in allfl], ne, e0, d)
    my = plans[rank]
    c_last = int(np.searchsorted(ts, e0 + my.hi * d, side="right")) - 1   # local close of my last edge
    send_h, recv_h = halo_lengths(comm, n, c_last)
    assert recv_h <= HALO
    comm.neighbour_exchange([t[HALO + c_last: HALO + n] for t in tens] if send_h else [],
                            [t[HALO - recv_h: HALO] for t in tens] if recv_h else [])
    ets, epx, eam, esd = (b[HALO - recv_h:] for b in back)                # extended shard
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
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sparse", [(2, False), (3, False), (2, True)])
def test_sharded_time_bars_match_single_process(tmp_path, orc, world, sparse):
    import torch.multiprocessing as mp
    gap = orc.SPARSE_GAP_MOD if sparse else orc.DENSE_GAP_MOD
    mp.spawn(_worker, args=(world, _free_port(), gap, str(tmp_path)), nprocs=world, join=True)
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
