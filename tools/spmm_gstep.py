#!/usr/bin/env python3
"""k_spmm_pipe: time against groups per workgroup (BICG_TEST=spmm-gstep, thousandths); default = cluster distance / m"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
X = np.random.default_rng(0).standard_normal((16, A.rows))
sg = (np.arange(16) + 1.0) * 0.01 / 16
for step in (None, 12225, 13000, 13430, 13700, 14000, 17900, 26850, 53700):
    if step is None:
        os.environ.pop("BICG_TEST", None)
    else:
        os.environ["BICG_TEST"] = f"spmm-gstep={step}"
    ctx = H.Context(H.single_rank_blocks(A))
    ms = min(ctx.spmm(X, sg)[1] for _ in range(6))
    print(f"gstep {step}: event-bracketed {1e3 * ms:.1f} us (kind {int(H.lib().bicg_last_spmm_windowed(ctx.h))})", flush=True)
    ctx.close()
