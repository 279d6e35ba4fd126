#!/usr/bin/env python3
"""A/B of the SpMV kernel variants on the bench workload (BICG_SPMV_VARIANT bits: 1 XCD, 2 wide, 4 nt)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
b = ctx.spmv(np.ones(A.rows))
ms = min(ctx.spmv_bench(200) for _ in range(3))
ctx.load(np.zeros(A.rows), b)
ctx.run_begin("bicgstab", tol=0.0, max_iter=220, check_every=200)
ctx.run_iterate(20)
import time; t=time.perf_counter(); ctx.run_iterate(200); dt=time.perf_counter()-t
r = ctx.run_end()
print("variant", os.environ.get("BICG_SPMV_VARIANT"), "spmv_us %%.1f" %% (ms*1e3), "GB/s %%.0f" %% ((12*A.nnz+4*(A.rows+1)+16*A.rows)/ms/1e6), "iter_us %%.1f" %% (dt/200*1e6), "relres %%.3e" %% np.sqrt(r.dot_r/r.dot_zero), "bsum %%.17g" %% b.sum())
''' % ROOT
for v in sys.argv[1:] or ["0", "1", "2", "3", "4", "5", "6", "7"]:
    env = dict(os.environ, BICG_SPMV_VARIANT=v)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-500:], flush=True)
