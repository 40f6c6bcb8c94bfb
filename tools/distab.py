#!/usr/bin/env python3
"""Host time of the pieces of one sharded time-bar step (self-loop communicator, 1 GPU): how long the host takes to ENQUEUE the
exchange, the interior bars, the wait and the boundary bar, against the step's wall time.  usage: distab.py [ticks] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finmlkit_amd import _ffi, engine
from finmlkit_amd.dist import Comm, ShardedTimeBars
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _ffi.default_context()
trades = engine.DeviceTrades.synth(n, seed=1, first=0, ctx=ctx)
comm = Comm(ctx, 0, 1, f"/dev/shm/fmk_distab_{os.getpid()}", "rccl", self_loop=True)
sh = ShardedTimeBars(trades, 0, 1, 60.0, True, self_loop=True).setup(comm)
for _ in range(3):
    sh.step(comm)
ctx.sync(); comm.sync()
acc = [0.0] * 4
import ctypes as C
ctx.call("fmk_profile_enable", C.c_int(1))
ctx.set_enqueue_only(True)
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter(); comm.exchange(sh.send_slices(), sh.recv_slices())
    b = time.perf_counter(); sh.enqueue_interior()
    c = time.perf_counter(); comm.wait()
    d = time.perf_counter(); sh.enqueue_boundary()
    e = time.perf_counter()
    for i, v in enumerate((b - a, c - b, d - c, e - d)):
        acc[i] += v
t1 = time.perf_counter()
comm.sync(); ctx.sync()
t2 = time.perf_counter()
kms = (C.c_double * 256)(); kn = C.c_int()
ctx.call("fmk_profile_read", kms, C.c_int(256), C.byref(kn))
kern = sum(kms[i] for i in range(kn.value)) / steps
print(f"dominant kernel {kern:.3f} ms/step, step - kernel {(t2-t0)/steps*1e3 - kern:.3f} ms")
print(f"one_call={sh.one_call} eo_census={os.environ.get('FMK_TB_PIPE_EO_CENSUS','1')}: host enqueue per step "
      f"exchange {acc[0]/steps*1e3:.3f} interior {acc[1]/steps*1e3:.3f} wait {acc[2]/steps*1e3:.3f} boundary {acc[3]/steps*1e3:.3f} ms; "
      f"enqueue loop {(t1-t0)/steps*1e3:.3f} ms/step, with the final sync {(t2-t0)/steps*1e3:.3f} ms/step")
