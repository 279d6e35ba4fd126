#!/usr/bin/env python3
"""20 back-to-back SpMVs on the bench matrix (for rocprofv3 counter passes); SPMV_KIND=fem_like: the FEM-like matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
n = int(os.environ.get("SPMV_N", synth.TRANSPORT_N))
A = synth.fem_like(n, scale_decades=2.0) if os.environ.get("SPMV_KIND", "transport") == "fem_like" else synth.transport_like(n=n, scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
print(ctx.plan_info(), "ms per SpMV", ctx.spmv_bench(20))
