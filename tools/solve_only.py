#!/usr/bin/env python3
"""A short solve on the bench matrix with the library's environment knobs as set by the caller (for rocprofv3 passes):
   SOLVE_METHOD (bicgstab), SOLVE_ITERS (30), SOLVE_N (1602111)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(n=int(os.environ.get("SOLVE_N", synth.TRANSPORT_N)), scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
b = ctx.spmv(np.ones(A.rows))
k = int(os.environ.get("SOLVE_ITERS", "30"))
res = ctx.solve(os.environ.get("SOLVE_METHOD", "bicgstab"), b, tol=0.0, max_iter=k, check_every=k)
print("iterations", res["k"], "relres", float(np.sqrt(res["dot_r"] / res["dot_zero"])))
