import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
b = ctx.spmv(np.ones(A.rows))
sigma = (np.arange(16) + 1.0) * 0.01 / 16
out = []
for which in ("shifted_lopbicgstab", "shifted_pipe_lopbicgstab"):
    best = 1e9
    for rep in range(3):
        rs = ctx.solve_shifted(b + sigma[7], sigma, 7, tol=0.0, max_iter=60, check_every=60, which=which)
        best = min(best, 1e6 * rs["result"].seconds / 60)
    out.append(f"{which} {best:.1f}us")
for m in ("ca_bicgstab", "pipe_bicgstab"):
    best = 1e9
    for rep in range(3):
        ctx.load(np.zeros(A.rows), b); ctx.run_begin(m, tol=0.0, max_iter=120, check_every=100); ctx.run_iterate(20)
        t = time.perf_counter(); ctx.run_iterate(100); best = min(best, (time.perf_counter() - t) / 100 * 1e6); ctx.run_end()
    out.append(f"{m} {best:.1f}us")
print({k: os.environ.get(k) for k in ("BICG_X_NT", "BICG_SET_NT")}, "  ".join(out))
