#!/usr/bin/env python3
"""16 shifts on the rank one of 8 GPUs holds (200 264 rows of the Transport-shaped matrix): microseconds per iteration of
shifted_lopbicgstab and shifted_pipe_lopbicgstab in the multi-launch form and (pipelined) as persistent launches.
   python tools/shifted_small_rank.py [rows] [nshifts]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mpi_bicgstab_amd import hipsolver as H, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else (synth.TRANSPORT_N + 7) // 8
nsh = int(sys.argv[2]) if len(sys.argv) > 2 else 16
seed = nsh // 2 - 1
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(n=rows, scale_decades=2.0)
sigma = (np.arange(nsh) + 1.0) * 0.01 / nsh
ones = np.ones(A.rows)
for env, label in (("1", "persistent"), ("0", "multi-launch")):
    H.switches(persist_shifted=env)
    ctx = H.Context(H.single_rank_blocks(A))
    b = ctx.spmv(ones) + sigma[seed] * ones
    for which in ("shifted_pipe_lopbicgstab", "shifted_lopbicgstab"):
        best = None
        for its in (100, 500, 500):
            t = time.perf_counter()
            got = ctx.solve_shifted(b, sigma, seed, tol=0.0, max_iter=its, check_every=128, which=which)
            dt = time.perf_counter() - t
            sec = got["result"].seconds
            if its == 500: best = sec if best is None else min(best, sec)
        print(f"{rows} rows, {nsh} shifts, {which:28s} {label:12s} persistent launches: {ctx.last_shifted_persistent()}  "
              f"{1e6 * best / 500:7.2f} us per iteration (500 its, solve timer)", flush=True)
    ctx.close()
