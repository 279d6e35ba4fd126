#!/usr/bin/env python3
"""BASELINE.json configs[3] on one GPU (7-point Laplacian 512^3, generated and planned on the device): 25 iterations of plain and
CA-BiCGStab + 20 products back to back (for rocprofv3 kernel stats, and as a quick timing).   python tools/lap512_only.py [m]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx, nnz, ps, gs = H.Context.stencil7_on_device(m, synth.LAPLACE_WEIGHTS)
n = m ** 3
print(m, "plan %.3f s, generate %.3f s, uniform entries %d, constant entries %d of %d, masked rows %d, bytes one product streams from the matrix arrays %d, flags %s"
      % (ps, gs, ctx.uniform_entries(), ctx.constant_entries(), nnz, ctx.masked_rows(), ctx.spmv_matrix_bytes(), [k for k, v in ctx.flags().items() if v]), flush=True)
b = ctx.spmv(np.ones(n))
for method in ("bicgstab", "ca_bicgstab"):
    ctx.load(np.zeros(n), b)
    ctx.run_begin(method, tol=0.0, max_iter=25, check_every=25)
    ctx.run_iterate(5); ctx.sync()
    t = time.perf_counter(); ctx.run_iterate(20); ctx.sync(); dt = (time.perf_counter() - t) / 20
    ctx.run_end()
    print(m, method, "%.3f ms per iteration" % (dt * 1e3), flush=True)
print(m, "spmv back to back %.3f ms" % ctx.spmv_bench(20), flush=True)
ctx.close()
