import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.fem_like()
lens = np.diff(A.ptr.astype(np.int64))
print("fem_like nnz", A.nnz, "mean", lens.mean(), "min", lens.min(), "max", lens.max(), flush=True)
bytes_spmv = 12 * A.nnz + 4 * (A.rows + 1) + 16 * A.rows
for env in ({}, {"BICG_NO_SELL": "1"}):
    os.environ.pop("BICG_NO_SELL", None); os.environ.update(env)
    ctx = H.Context(H.single_rank_blocks(A))
    ms = min(ctx.spmv_bench(50) for _ in range(3))
    print(env, ctx.plan_info(), f"spmv {ms*1e3:.1f} us {bytes_spmv/ms/1e6:.0f} GB/s", flush=True)
    ctx.close()
