#!/usr/bin/env python3
"""Is a slow variant leg of bench.py a first-use effect? Every method timed 3 x with K = 20 / W = 5 and with K = 200."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
b = ctx.spmv(np.ones(A.rows)); x0 = np.zeros(A.rows)
def timed(m, K, W, ce=None):
    ctx.load(x0, b)
    ctx.run_begin(m, tol=0.0, max_iter=W + K, check_every=ce or max(W, K, 1), krr=50, nrr=2)
    ctx.run_iterate(W); ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ctx.run_iterate(K); ctx.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.run_end()
    return 1e3 * dt / K
for K, W in ((20, 5), (200, 20)):
    for rep in range(3):
        print(K, rep, {m: round(timed(m, K, W), 4) for m in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr")}, flush=True)
print("check_every 16:", {m: round(timed(m, 200, 20, 16), 4) for m in ("bicgstab", "ca_bicgstab", "pipe_bicgstab")})
