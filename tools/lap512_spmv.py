#!/usr/bin/env python3
"""10 products back to back on the 512^3 Laplacian planned on the device (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx, nnz, ps, gs = H.Context.stencil7_on_device(m, synth.LAPLACE_WEIGHTS)
print(m, "ms per product", ctx.spmv_bench(10))
ctx.close()
