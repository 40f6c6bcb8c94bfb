"""GPU: the sharded time-bar step of finmlkit_amd/dist.py (what bench.py runs at --gpus N > 1).

* W *virtual ranks* on ONE device in one process: the halo travels by a device copy, everything else -- global clock,
  edge plan, local index, interior bars, boundary bar from [halo | head of the shard] -- is the code the real ranks run.
* W = 2 REAL processes on one device through the C entry points `fmk_comm_*` with the host-staged transport
  (`ShardedTimeBars.setup(comm)` / `.step(comm)`, exactly bench.py's calls).
* the RCCL transport with one rank that is its own neighbour (ncclSend / ncclRecv to self): librccl is loaded, a
  communicator is built, the grouped exchange runs on the communicator's stream and the event ordering is exercised.
The concatenated per-rank outputs must equal one un-sharded run, bit for bit."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

KEYS = ["open", "high", "low", "close", "volume", "vwap", "trades", "median_trade_size"]


def _copy_halo(ctx, left, right):
    """What the exchange does between two virtual ranks: left.send_slices() -> right.recv_slices()."""
    for (sp, sb), (rp, rb) in zip(left.send_slices(), right.recv_slices()):
        assert sb == rb and sb > 0
        ctx.call("fmk_d2d", C.c_void_p(rp), C.c_void_p(sp), C.c_size_t(sb))


def _setup_virtual(dist, shards, interval, want_median=True, with_side=False):
    world = len(shards)
    ranks = [dist.ShardedTimeBars(t, r, world, interval, want_median, with_side=with_side) for r, t in enumerate(shards)]
    spans = [list(s.span()) for s in ranks]                    # all-gather #1
    send_h = [s.make_plan(spans) for s in ranks]               # all-gather #2
    assert send_h[-1] == 0
    for r, s in enumerate(ranks):
        s.set_halo(send_h[r - 1] if r else 0)
    return ranks


def _run_sharded(engine, dist, ctx, world, n, gap_mod, interval, want_median=True, steps=1):
    shards = [engine.DeviceTrades.synth(n, seed=42, first=r * n, gap_mod=gap_mod, ctx=ctx) for r in range(world)]
    ranks = _setup_virtual(dist, shards, interval, want_median)
    for _ in range(steps):                                     # a second step reuses every buffer
        for r in range(1, world):
            _copy_halo(ctx, ranks[r - 1], ranks[r])
        for s in ranks:
            s.enqueue_interior()
        nb = [s.enqueue_boundary() for s in ranks]
    out = {k: np.concatenate([s.out[k].to_host()[:b] for s, b in zip(ranks, nb)]) for k in KEYS if want_median or k != KEYS[-1]}
    clock = np.concatenate([ranks[0].clock.to_host()[:1]] + [s.clock.to_host()[1:b + 1] for s, b in zip(ranks, nb)])
    return clock, out


@pytest.mark.parametrize("world,n,sparse,interval", [(2, 400_000, False, 60.0), (3, 250_000, False, 60.0),
                                                    (4, 100_000, False, 300.0), (2, 300_000, True, 60.0),
                                                    (8, 60_000, False, 60.0)])
def test_virtual_ranks_match_unsharded(orc, world, n, sparse, interval):
    from finmlkit_amd import _ffi, dist, engine
    ctx = _ffi.default_context()
    gap = engine.SPARSE_GAP_MOD if sparse else engine.DENSE_GAP_MOD
    clock, got = _run_sharded(engine, dist, ctx, world, n, gap, interval, steps=2)
    whole = engine.DeviceTrades.synth(world * n, seed=42, gap_mod=gap, ctx=ctx)
    wclock, wci = whole.time_bar_index(interval)
    want = engine.to_host(whole.bar_ohlcv(wci))
    np.testing.assert_array_equal(clock, wclock.to_host())
    for k in KEYS:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)   # same kernels, same ticks -> identical
    # and against the oracle (vwap: tree vs sequential order)
    ts, px, am, sd = orc.synth(42, 0, world * n, gap)
    oclk, oci = orc._time_bar_indexer(ts, interval)
    np.testing.assert_array_equal(clock, oclk)
    for k, w in zip(KEYS, orc.comp_bar_ohlcv(px, am, oci)):
        if k == "vwap":
            np.testing.assert_allclose(got[k], w, rtol=1e-9)
        else:
            np.testing.assert_array_equal(got[k], w, err_msg=k)


def test_shard_without_complete_bar_is_rejected():
    from finmlkit_amd import _ffi, dist, engine
    ctx = _ffi.default_context()
    shards = [engine.DeviceTrades.synth(500, seed=1, first=r * 500, ctx=ctx) for r in range(2)]
    ranks = [dist.ShardedTimeBars(t, r, 2, 3600.0) for r, t in enumerate(shards)]
    spans = [list(s.span()) for s in ranks]
    with pytest.raises(ValueError, match="complete bar close"):
        ranks[1].make_plan(spans)


@pytest.mark.parametrize("world,n,interval", [(3, 250_000, 60.0), (2, 300_000, 7200.0)])
def test_virtual_ranks_features_match_unsharded(world, n, interval):
    """cfg 4 on shards: order-flow + footprints of every rank's bars == the un-sharded run (same kernels, same ticks)."""
    from finmlkit_amd import _ffi, dist, engine
    ctx = _ffi.default_context()
    shards = [engine.DeviceTrades.synth(n, seed=42, first=r * n, ctx=ctx) for r in range(world)]
    ranks = _setup_virtual(dist, shards, interval, True, with_side=True)
    for r in range(1, world):
        _copy_halo(ctx, ranks[r - 1], ranks[r])
    for s in ranks:
        s.enqueue_interior()
        s.enqueue_boundary()
    parts = [s.features(0.01, 3.0) for s in ranks]
    whole = engine.DeviceTrades.synth(world * n, seed=42, ctx=ctx)
    _, wci = whole.time_bar_index(interval)
    o, d, nz, off, flat, bar, bad = whole.bars_fused(wci, 0.01, 3.0, want_median=False)
    wd, wflat, wbar = engine.to_host(d), engine.to_host(flat), engine.to_host(bar)
    for k, w in wd.items():
        got = np.concatenate([p[0][k] for p in parts])
        if k in ("mean_spread", "max_spread"):        # bar 0: wrap-around tick of the array at hand (reference quirk)
            got, w = got[1:], w[1:]
        np.testing.assert_array_equal(got, w, err_msg=k)
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), np.diff(off.to_host()))
    for k, w in wflat.items():
        np.testing.assert_array_equal(np.concatenate([p[2][k] for p in parts]), w, err_msg=k)
    for k, w in wbar.items():
        np.testing.assert_array_equal(np.concatenate([p[3][k] for p in parts]), w, err_msg=k)


@pytest.mark.parametrize("world,n,window,half_life,mean0", [(3, 300_000, 5.0, 60.0, False), (4, 150_000, 60.0, 5.0, True),
                                                          (2, 200_001, 0.5, 600.0, False)])
def test_virtual_ranks_tick_level_features(world, n, window, half_life, mean0):
    HALO = 1 << 16
    """Sharded comp_lagged_returns (raw-tick halo of one window) and ewmst (maps of the lower ranks -> incoming state)
    == the un-sharded run: returns bit-identical, sigma within the float tolerance (different composition order)."""
    from finmlkit_amd import _ffi, dist, engine
    ctx = _ffi.default_context()
    shards = [engine.DeviceTrades.synth(n, seed=42, first=r * n, ctx=ctx, headroom=HALO) for r in range(world)]
    tl = [dist.ShardedTickLevel(t, r, world) for r, t in enumerate(shards)]
    firsts = [t.first_last_ts()[0] for t in shards]
    start = [tl[r].returns_send_start(firsts[r + 1], window) for r in range(world - 1)]
    recv = [0] + [n - s for s in start]
    for r in range(1, world):
        h = recv[r]
        assert 1 <= h <= HALO
        for src, dst in zip(shards[r - 1]._backing, shards[r]._backing):
            s_, d_ = src.view(HALO + start[r - 1], h), dst.view(HALO - h, h)
            ctx.call("fmk_d2d", d_.p, s_.p, C.c_size_t(s_.nbytes))
    r_ext = [tl[r].lagged_returns(recv[r], window, True) for r in range(world)]
    whole = engine.DeviceTrades.synth(world * n, seed=42, ctx=ctx)
    wr = whole.lagged_returns(window, True)
    want_r = wr.to_host()
    for r in range(world):
        np.testing.assert_array_equal(r_ext[r].to_host()[recv[r]:], want_r[r * n:(r + 1) * n], err_msg=f"returns rank {r}")
    for r in range(1, world):                                   # the left neighbour's last return (8 bytes)
        tl[r].set_left_value(r_ext[r], recv[r], float(r_ext[r - 1].view(r_ext[r - 1].n - 1, 1).to_host()[0]))
    maps = [tl[r].ewmst_map(r_ext[r], recv[r], half_life, mean0) for r in range(world)]
    want_s = whole.ewmst(wr, half_life, mean0=mean0).to_host()
    for r in range(world):
        got = tl[r].ewmst(r_ext[r], recv[r], maps[:r], half_life, mean0=mean0).to_host()
        w = want_s[r * n:(r + 1) * n]
        assert got.shape == w.shape and np.array_equal(np.isnan(got), np.isnan(w)), f"rank {r}"
        np.testing.assert_allclose(got, w, rtol=1e-9, atol=0, equal_nan=True, err_msg=f"sigma rank {r}")


# ---- real processes, real C entry points ----------------------------------------------------------------------------
def _proc_worker(rank, world, path, n, interval, out_dir, transport, steps):
    os.environ["FMK_DEVICE"] = "0"                              # both ranks share the box's one GPU
    from finmlkit_amd import _ffi, dist, engine
    ctx = _ffi.default_context()
    comm = dist.Comm(ctx, rank, world, path, transport, self_loop=(world == 1), ring_bytes=8192, timeout_s=120.0)
    t = engine.DeviceTrades.synth(n, seed=42, first=rank * n, ctx=ctx)
    shard = dist.ShardedTimeBars(t, rank, world, interval, True, self_loop=(world == 1)).setup(comm)
    for _ in range(steps):
        nb = shard.step(comm)
    ctx.sync()
    comm.sync()
    res = {k: v.to_host()[:nb] for k, v in shard.out.items()}
    res["clock"] = shard.clock.to_host()[:nb + 1]
    if world == 1:                                              # the self-loop's extra bar sits in the spare slot
        res["loop_trades"] = shard._out["trades"].view(nb, 1).to_host()
        res["loop_expect"] = np.array([shard.recv_h - 1 + shard._head], dtype=np.int64)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    comm.barrier()
    comm.close()


def _spawn(world, args):
    mpx = mp.get_context("spawn")
    procs = [mpx.Process(target=_proc_worker, args=(r, world) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("worker hung")
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"


def _whole(world, n, interval):
    from finmlkit_amd import _ffi, engine
    ctx = _ffi.default_context()
    whole = engine.DeviceTrades.synth(world * n, seed=42, ctx=ctx)
    wclock, wci = whole.time_bar_index(interval)
    return wclock.to_host(), engine.to_host(whole.bar_ohlcv(wci))


@pytest.mark.parametrize("world,n,interval", [(2, 500_000, 60.0), (3, 200_000, 60.0)])
def test_processes_host_transport_match_unsharded(tmp_path, world, n, interval):
    _spawn(world, (str(tmp_path / "rdv"), n, interval, str(tmp_path), "host", 3))
    parts = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(world)]
    wclock, want = _whole(world, n, interval)
    clock = np.concatenate([parts[0]["clock"][:1]] + [p["clock"][1:] for p in parts])
    np.testing.assert_array_equal(clock, wclock)
    for k in KEYS:
        np.testing.assert_array_equal(np.concatenate([p[k] for p in parts]), want[k], err_msg=k)


def test_rccl_self_loop_one_rank(tmp_path):
    """librccl behind the C ABI: communicator of size 1, ncclSend/ncclRecv to self on the communicator's stream, event
    ordering against the context's stream; the rank's own bars are untouched and equal the un-sharded run, the extra
    boundary bar made of [own tail | own head] has exactly the ticks it was given."""
    n, interval = 600_000, 60.0
    _spawn(1, (str(tmp_path / "rdv"), n, interval, str(tmp_path), "rccl", 4))
    p = dict(np.load(tmp_path / "rank0.npz"))
    wclock, want = _whole(1, n, interval)
    np.testing.assert_array_equal(p["clock"], wclock)
    for k in KEYS:
        np.testing.assert_array_equal(p[k], want[k], err_msg=k)
    np.testing.assert_array_equal(p["loop_trades"], p["loop_expect"])


def _rccl_pair_worker(rank, world, path, out_dir):
    os.environ["FMK_DEVICE"] = "0"                              # two ranks on ONE device: RCCL refuses (or not) -- for both
    from finmlkit_amd import _ffi, dist
    ctx = _ffi.default_context()
    outcome = "ok"
    try:
        comm = dist.Comm(ctx, rank, world, path, "rccl", ring_bytes=8192, timeout_s=25.0)
        comm.barrier()
        comm.close()
    except _ffi.FmkError as e:
        outcome = f"error: {e}"
        # what bench.py does next: the same step over the host-staged transport, on every rank
        comm = dist.Comm(ctx, rank, world, path + ".host", "host", ring_bytes=8192, timeout_s=25.0)
        comm.barrier()
        comm.close()
    with open(os.path.join(out_dir, f"outcome{rank}.txt"), "w") as f:
        f.write(outcome)


def test_rccl_two_ranks_on_one_device_agree_on_the_outcome(tmp_path):
    """Two processes ask for the RCCL transport on the box's single GPU.  Whatever librccl makes of that (it normally rejects
    two ranks on one device), BOTH ranks must come out of fmk_comm_create the same way and within the deadline -- communicator
    creation and the first exchange run under a watchdog and the ranks agree on the result before anyone returns -- so that
    the host-staged fallback of bench.py is taken by all ranks or by none."""
    mpx = mp.get_context("spawn")
    procs = [mpx.Process(target=_rccl_pair_worker, args=(r, 2, str(tmp_path / "rdv"), str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank hung in the RCCL rendezvous")
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    outcomes = [open(tmp_path / f"outcome{r}.txt").read() for r in range(2)]
    print("outcomes:", outcomes)
    assert outcomes[0].startswith("ok") == outcomes[1].startswith("ok")


def _run_bench(extra, env_extra, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "FMK_BENCH_RDV", "FMK_DEVICE")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, lines, json


def test_plain_bench_two_ranks_spawns_itself(tmp_path):
    """`python3 bench.py --gpus 2` with NO launcher: the process becomes rank 0, starts rank 1 itself, the
    two meet in the rendezvous file and run the sharded step; rc 0 and exactly ONE JSON line on stdout, with the transport,
    the per-rank step times and the exchange's own time in it.  FMK_BENCH_ONE_DEVICE puts both ranks on the box's single GPU,
    which asks for the host-staged transport (RCCL refuses two ranks on one device)."""
    r, lines, json = _run_bench(["--gpus", "2", "--ticks", "20000000", "--steps", "3", "--warmup", "1"],
                                {"FMK_BENCH_ONE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["transport"] == "host" and line["config"]["launcher"] == "self-spawned ranks"
    assert "valid" not in line
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["exchange_ms"]) == 2 and len(pr["dominant_kernel_ms"]) == 2
    assert pr["ms_per_step_min"] <= pr["ms_per_step_max"] == max(pr["ms_per_step"])
    assert abs(line["ms_per_step"] - pr["ms_per_step_max"]) < 1e-9
    assert line["value"] == pytest.approx(2 * 20_000_000 * 3 / (line["ms_per_step"] * 3e-3), rel=1e-6)
    assert "cpu_baseline" not in line                                     # rank 0 at N = 1 only


def test_plain_bench_eight_ranks_one_device(tmp_path):
    """The shape of the driver's N = 8 run, on the box's one GPU (FMK_BENCH_ONE_DEVICE: host-staged transport): eight processes, one
    JSON line, per-rank arrays of length 8, the job's bar count equal to the un-sharded run's, and every rank's own log file."""
    from finmlkit_amd import _ffi, engine
    n, world = 2_000_000, 8
    r, lines, json = _run_bench(["--gpus", str(world), "--ticks", str(n), "--steps", "3", "--warmup", "1"],
                                {"FMK_BENCH_ONE_DEVICE": "1", "FMK_BENCH_LOGDIR": str(tmp_path)}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["transport"] == "host" and "valid" not in line
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == world and len(pr["exchange_ms"]) == world and len(pr["dominant_kernel_ms"]) == world
    assert abs(line["ms_per_step"] - max(pr["ms_per_step"])) < 1e-9
    assert line["value"] == pytest.approx(world * n * 3 / (line["ms_per_step"] * 3e-3), rel=1e-6)
    whole = engine.DeviceTrades.synth(world * n, seed=42, ctx=_ffi.default_context())
    _, wci = whole.time_bar_index(60.0)
    assert line["config"]["n_bars_total"] == wci.n - 1                     # the stitched shards have the bars of one stream
    for k in range(world):
        text = (tmp_path / f"bench_rank{k}.log").read_text()
        assert f"rank {k} of {world}" in text and "communicator up, transport host" in text
        assert f"[selftest] rank {k} of {world}" in text and "librccl: " in text          # the first-contact report comes first


def test_bench_eight_ranks_rccl_failure_ends_every_rank(tmp_path):
    """Eight ranks on ONE device asking for RCCL: librccl refuses, every rank learns it inside the deadline, the step runs
    host-staged, the run ends with rc 3 and no JSON on stdout -- and each rank's log says why (the 2-rank form is the test below)."""
    r, lines, json = _run_bench(["--gpus", "8", "--ticks", "2000000", "--steps", "2", "--warmup", "1"],
                                {"FMK_BENCH_SAME_DEVICE_RCCL": "1", "FMK_BENCH_LOGDIR": str(tmp_path)}, timeout=900)
    if r.returncode == 0:                                                 # this librccl accepted eight ranks on one device
        assert json.loads(lines[0])["config"]["transport"] == "rccl"
        return
    assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
    assert not [ln for ln in lines if ln.lstrip().startswith("{")], lines
    for k in range(8):
        assert "host-staged fallback" in (tmp_path / f"bench_rank{k}.log").read_text()


def test_bench_rccl_fallback_is_not_a_result(tmp_path):
    """Two ranks on ONE device asking for RCCL (developer switch FMK_BENCH_SAME_DEVICE_RCCL): librccl refuses the communicator,
    every rank learns it, the step still runs host-staged -- but the run ends with rc 3 and NOTHING on stdout: a scaling curve
    cannot silently be a host-staged one."""
    r, lines, json = _run_bench(["--gpus", "2", "--ticks", "5000000", "--steps", "2", "--warmup", "1"],
                                {"FMK_BENCH_SAME_DEVICE_RCCL": "1"})
    if r.returncode == 0:                                                 # this librccl accepted two ranks on one device
        assert json.loads(lines[0])["config"]["transport"] == "rccl"
        return
    assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
    assert not [ln for ln in lines if ln.lstrip().startswith("{")], lines      # (librccl's version banner may be there)
    assert "host-staged fallback" in r.stderr and '"valid": false' in r.stderr


def test_bench_force_dist_reports_transport_and_exchange_time():
    r, lines, json = _run_bench(["--ticks", "20000000", "--steps", "3", "--warmup", "1", "--force-dist", "--no-extras",
                                 "--cpu-sample", "0"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(lines[-1])
    assert line["config"]["transport"] == "rccl" and line["n_gpus"] == 1
    assert len(line["per_rank"]["exchange_ms"]) == 1 and line["per_rank"]["exchange_ms"][0] > 0.0


def test_bench_single_gpu_line_has_transport_none():
    r, lines, json = _run_bench(["--ticks", "20000000", "--steps", "3", "--warmup", "1", "--no-extras", "--cpu-sample", "0"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["config"]["transport"] == "none" and line["n_gpus"] == 1 and "per_rank" not in line


# ---- first-contact kit (finmlkit_amd/dist.py: selftest) -------------------------------------------
def _run_selftest(env_extra, args=(), timeout=300):
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, "-m", "finmlkit_amd.dist", "--selftest", *args], cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_selftest_one_rank_reports_the_node_and_talks_to_itself_over_rccl(tmp_path):
    """`python -m finmlkit_amd.dist --selftest` with no launcher: the node report (devices, librccl path + version, the IPC
    environment) and a 1 KiB ncclSend / ncclRecv to itself on the communicator's stream, content checked: rc 0."""
    r = _run_selftest({"FMK_BENCH_RDV": str(tmp_path / "rdv")})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "devices: " in r.stdout and "gfx950" in r.stdout and "librccl: " in r.stdout and "ncclGetVersion" in r.stdout
    assert "HSA_ENABLE_IPC_MODE_LEGACY" in r.stdout
    assert "RCCL exchange of 1024 B to rank + 1: ok" in r.stdout and "SELFTEST rank 0/1: RCCL ok" in r.stdout


def test_selftest_two_ranks_on_one_device_say_why_rccl_fails_and_that_the_flow_works(tmp_path):
    """Two ranks on the box's ONE device: librccl refuses the communicator -- both ranks learn it inside the deadline, say why, run the
    same exchange host-staged and end with rc 3 (a librccl that accepts two ranks on one device ends with rc 0: also fine)."""
    import subprocess
    import sys
    procs = []
    for k in range(2):
        env = dict(os.environ, RANK=str(k), WORLD_SIZE="2", LOCAL_RANK=str(k), FMK_BENCH_RDV=str(tmp_path / "rdv"))
        procs.append(subprocess.Popen([sys.executable, "-m", "finmlkit_amd.dist", "--selftest", "--one-device", "--deadline", "20"],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    rcs = [p.returncode for p in procs]
    assert rcs[0] == rcs[1] and rcs[0] in (0, 3), (rcs, outs[0][-2000:], outs[1][-2000:])
    for k, out in enumerate(outs):
        assert f"[selftest] rank {k} of 2" in out and "librccl: " in out
        if rcs[0] == 3:
            assert "RCCL leg FAILED" in out and "host-staged: ok" in out and f"SELFTEST rank {k}/2: RCCL FAILED" in out


def _features_worker(rank, world, path, n, interval, out_dir):
    os.environ["FMK_DEVICE"] = "0"
    from finmlkit_amd import _ffi, dist, engine
    ctx = _ffi.default_context()
    comm = dist.Comm(ctx, rank, world, path, "host", ring_bytes=1 << 16, timeout_s=120.0)
    t = engine.DeviceTrades.synth(n, seed=42, first=rank * n, ctx=ctx)
    shard = dist.ShardedTimeBars(t, rank, world, interval, False, with_side=True).setup(comm)
    shard.step(comm)
    ctx.sync()
    comm.sync()
    d, lv, flat, bar = shard.features(0.01, 3.0)
    np.savez(os.path.join(out_dir, f"feat{rank}.npz"), lv=lv, **{"d_" + k: v for k, v in d.items()},
             **{"f_" + k: v for k, v in flat.items()}, **{"b_" + k: v for k, v in bar.items()})
    comm.barrier()
    comm.close()


def test_eight_processes_cfg4_features_across_shard_boundaries(tmp_path):
    """cfg 4 (order flow + footprints) on EIGHT real processes sharing the box's GPU, host-staged transport: every rank's bars --
    the boundary bar stitched from [halo | head] included -- equal the un-sharded run's, bit for bit (the spread columns of the
    stream's very first bar excepted: the reference's wrap-around tick is 'the last tick of the array at hand')."""
    from finmlkit_amd import _ffi, engine
    world, n, interval = 8, 150_000, 60.0
    mpx = mp.get_context("spawn")
    procs = [mpx.Process(target=_features_worker, args=(r, world, str(tmp_path / "rdv"), n, interval, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("worker hung")
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    parts = [dict(np.load(tmp_path / f"feat{r}.npz")) for r in range(world)]
    ctx = _ffi.default_context()
    whole = engine.DeviceTrades.synth(world * n, seed=42, ctx=ctx)
    _, wci = whole.time_bar_index(interval)
    o, d, nz, off, flat, bar, bad = whole.bars_fused(wci, 0.01, 3.0, want_median=False)
    for k, w in engine.to_host(d).items():
        got = np.concatenate([p["d_" + k] for p in parts])
        if k in ("mean_spread", "max_spread"):
            got, w = got[1:], w[1:]
        np.testing.assert_array_equal(got, w, err_msg=k)
    np.testing.assert_array_equal(np.concatenate([p["lv"] for p in parts]), np.diff(off.to_host()))
    for k, w in engine.to_host(flat).items():
        np.testing.assert_array_equal(np.concatenate([p["f_" + k] for p in parts]), w, err_msg=k)
    for k, w in engine.to_host(bar).items():
        np.testing.assert_array_equal(np.concatenate([p["b_" + k] for p in parts]), w, err_msg=k)
