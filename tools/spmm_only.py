#!/usr/bin/env python3
"""SpMM of 16 vectors on the bench matrix: the pipelined kernel (k_spmm_pipe) against the windowed one (BICG_PLAN=spmm-window=1),
columns compared bit for bit with each other and with 16 single products; a few more launches for rocprofv3 kernel stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
kind = os.environ.get("SPMM_MATRIX", "transport")          # transport | fem_like | mesh_rcm | mesh_generator
if kind.startswith("mesh_"):
    from mpi_bicgstab_amd import mesh
    import tempfile
    cache = os.path.join(tempfile.gettempdir(), "bicg_mesh_cache")      # (bench.py's: generating the mesh under rocprofv3 takes minutes)
    os.makedirs(cache, exist_ok=True)
    A = mesh.fem_unstructured(117, kind[5:], 2.0, cache_dir=cache)
elif kind == "fem_like":
    A = synth.fem_like(scale_decades=2.0)
else:
    A = synth.transport_like(scale_decades=2.0)
print("matrix", kind, A.rows, "rows", A.nnz if hasattr(A, "nnz") else len(A.val), "non-zeros", flush=True)
X = np.random.default_rng(0).standard_normal((16, A.rows))
sg = (np.arange(16) + 1.0) * 0.01 / 16
out = {}
for name, tok in (("windowed", 1), ("pipelined", 3)):
    H.switches(spmm_window=tok)
    ctx = H.Context(H.single_rank_blocks(A))
    for _ in range(5):
        Y, ms = ctx.spmm(X, sg)
    out[name] = Y
    print(name, "kind", int(H.lib().bicg_last_spmm_windowed(ctx.h)), "spmm ms", round(ms, 4), "spmv ms", round(ctx.spmv_bench(50), 4), flush=True)
    if name == "pipelined":
        y3 = ctx.spmv(X[3]) + sg[3] * X[3]
        print("column 3 equals the single product + shift:", bool(np.array_equal(Y[3], y3)))
        Y5, _ = ctx.spmm(X[:5], sg[:5])
        print("5 vectors:", bool(np.array_equal(Y5, Y[:5])))
        b = np.random.default_rng(1).standard_normal(A.rows)
        r = ctx.shifted_residuals(X, b, sg)
        ref = np.array([np.linalg.norm(b - Y[j]) / np.linalg.norm(b) for j in range(16)])
        print("residual norms max rel diff", float(np.abs(np.asarray(r) - ref).max() / ref.max()))
    ctx.close()
print("bit-identical columns:", bool(np.array_equal(out["windowed"], out["pipelined"])))
