#!/usr/bin/env python3
"""Where the 10-80 ms "queue stall" of bench.py's variant legs comes from: the runtime loads a translation unit's code object at
the first look-up of one of its kernels. One process per setting (BICG_PRELOAD=0 / 1): bicg_create, then chunks of 10 iterations
of each solver, wall time per chunk -- a chunk that loads a code object stands out by 10-80 ms.
    BICG_PRELOAD=0 python tools/preload_check.py ; BICG_PRELOAD=1 python tools/preload_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
blocks = H.single_rank_blocks(A)
t = time.perf_counter(); ctx = H.Context(blocks); print("BICG_PRELOAD=%s  bicg_create %.1f ms" % (os.environ.get("BICG_PRELOAD", "1"), 1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); b = ctx.spmv(np.ones(A.rows)); print("first product (upload of x, launch, download) %.1f ms" % (1e3 * (time.perf_counter() - t)))
x0 = np.zeros(A.rows)
for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr", "bicgstab"):
    ctx.load(x0, b)
    ctx.run_begin(method, tol=0.0, max_iter=80, check_every=80, krr=50, nrr=2)
    ms = []
    for _ in range(8):
        t = time.perf_counter(); ctx.run_iterate(10); ctx.sync(); ms.append(1e3 * (time.perf_counter() - t))
    ctx.run_end()
    print("%-18s ms per chunk of 10 iterations: %s" % (method, " ".join("%6.2f" % v for v in ms)))
ctx.close()
