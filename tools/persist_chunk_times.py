"""us per iteration of the persistent forms on a 200 k-row rank as a function of the host check interval
(bicg_options.check_every = iterations per persistent launch): what the per-launch set-up costs the drop-in path."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_bicgstab_amd import hipsolver as H      # noqa: E402
from mpi_bicgstab_amd import synth               # noqa: E402

H.lib().bicg_comm_init_single(0)
A = synth.transport_like(200264, scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
assert ctx.flags()["persist"], ctx.flags()
b = ctx.spmv(np.ones(A.rows))
for method in ("pipe_bicgstab", "bicgstab", "ca_bicgstab"):
    for ce in (8, 16, 32, 64, 128, 512):
        best = 1e9
        for _ in range(3):
            r = ctx.solve(method, b, tol=0.0, max_iter=512, check_every=ce)
            best = min(best, r["result"].iter_seconds / r["k"])
        print(f"{method:14s} check_every {ce:4d}: {best * 1e6:6.2f} us per iteration")
ctx.close()
