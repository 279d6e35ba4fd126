#!/usr/bin/env python3
"""Quick on-GPU probe: SpMV bandwidth on the Transport-shaped matrix + one plain solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth

n = int(os.environ.get("PROBE_N", synth.TRANSPORT_N))
t = time.time(); A = synth.transport_like(n=n, diag_base=float(os.environ.get("PROBE_DIAG", "16"))); print("gen", time.time() - t, "s nnz", A.nnz, flush=True)
H.lib().bicg_comm_init_single(0)
t = time.time(); ctx = H.Context(H.single_rank_blocks(A)); print("create", time.time() - t, "s", ctx.plan_info(), flush=True)
bytes_spmv = 12 * A.nnz + 4 * (A.rows + 1) + 16 * A.rows
for reps in (20, 200):
    ms = ctx.spmv_bench(reps)
    print(f"spmv reps={reps}: {ms*1e3:.1f} us  {bytes_spmv/ms/1e6:.1f} GB/s  ({bytes_spmv/1e6:.1f} MB)", flush=True)
b = A.matvec(np.ones(A.rows))
for m in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"):
    res = ctx.solve(m, b, krr=10, nrr=3, max_iter=200)
    r = res["result"]
    print(m, "k", res["k"], "relres", np.sqrt(res["dot_r"] / res["dot_zero"]), "err", np.abs(res["x"] - 1).max(),
          f"iter {1e3*r.iter_seconds/max(res['k'],1):.4f} ms/it total {r.seconds:.4f}s", flush=True)
# fixed-iteration timing, tol=0
ctx.load(np.zeros(A.rows), b)
for m in ("bicgstab", "ca_bicgstab", "pipe_bicgstab"):
    ctx.load(np.zeros(A.rows), b)
    r = ctx.run(m, tol=0.0, max_iter=12, check_every=12)
    print("fixed12", m, r.iterations, f"{1e3*r.iter_seconds/12:.4f} ms/it", flush=True)
