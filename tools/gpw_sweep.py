import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
b = ctx.spmv(np.ones(A.rows))
ms = min(ctx.spmv_bench(100) for _ in range(2))
out = []
for m in ("bicgstab", "pipe_bicgstab"):
    ctx.load(np.zeros(A.rows), b)
    ctx.run_begin(m, tol=0.0, max_iter=220, check_every=200)
    ctx.run_iterate(20)
    t=time.perf_counter(); ctx.run_iterate(200); dt=time.perf_counter()-t
    ctx.run_end(); out.append("%%s %%.1f" %% (m, dt/200*1e6))
print("gpw", os.environ.get("BICG_SELL_GPW"), "gpw_dots", os.environ.get("BICG_SELL_GPW_DOTS"), "spmv_us %%.1f" %% (ms*1e3), " ".join(out))
''' % ROOT
for spec in sys.argv[1:]:
    g, gd = spec.split(",")
    env = dict(os.environ, BICG_SELL_GPW=g, BICG_SELL_GPW_DOTS=gd)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-400:], flush=True)
