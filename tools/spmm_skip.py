#!/usr/bin/env python3
"""where the windowed SpMM's time goes: the kernel with parts of it switched off (BICG_TEST=spmm-skip=n; results are wrong)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
X = np.random.default_rng(0).standard_normal((16, A.rows))
sg = (np.arange(16) + 1.0) * 0.01 / 16
for skip in (0, 1, 2, 3, 4, 7):
    os.environ["BICG_TEST"] = f"spmm-skip={skip}"
    ctx = H.Context(H.single_rank_blocks(A))
    ms = min(ctx.spmm(X, sg)[1] for _ in range(5))
    print(f"skip={skip} (1 no staging loads, 2 no products, 4 no row heads): {1e3 * ms:.1f} us", flush=True)
    ctx.close()
