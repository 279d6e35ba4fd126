#!/usr/bin/env python3
"""SpMM of 16 vectors on the bench matrix, a few times (for rocprofv3 kernel stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
X = np.random.default_rng(0).standard_normal((16, A.rows))
sg = (np.arange(16) + 1.0) * 0.01 / 16
for _ in range(5):
    Y, ms = ctx.spmm(X, sg)
print("spmm ms", ms, "spmv ms", ctx.spmv_bench(50))
b = np.random.default_rng(1).standard_normal(A.rows)
for _ in range(3):
    r = ctx.shifted_residuals(X, b, sg)
