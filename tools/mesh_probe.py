#!/usr/bin/env python3
"""The unstructured FEM matrix (mpi_bicgstab_amd.mesh) on the GPU in its three numberings: which product kernel it gets, the
window plan, product time back to back, ms per plain iteration, bit-exactness of y = A x against the oracle.
    python tools/mesh_probe.py [--m 117] [--numberings generator,rcm,random] [--plan TOKENS] [--no-oracle]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=117)
ap.add_argument("--numberings", default="generator,rcm,random")
ap.add_argument("--plan", default=None, help="BICG_PLAN tokens for every context")
ap.add_argument("--no-oracle", action="store_true")
ap.add_argument("--methods", default="bicgstab")
ap.add_argument("--steps", type=int, default=100)
a = ap.parse_args()
if a.plan is not None:
    os.environ["BICG_PLAN"] = a.plan

from mpi_bicgstab_amd import hipsolver as H, mesh  # noqa: E402

H.lib().bicg_comm_init_single(0)
stream = {k: max(H.stream_bench(k, 1 << 30, 10) for _ in range(2)) for k in ("copy", "triad")}
print("STREAM", {k: round(v) for k, v in stream.items()}, flush=True)
for kind in a.numberings.split(","):
    t0 = time.time()
    A = mesh.fem_unstructured(a.m, kind, scale_decades=2.0, cache_dir="/tmp")
    t_gen = time.time() - t0
    t0 = time.time()
    ctx = H.Context(H.single_rank_blocks(A))
    t_plan = time.time() - t0
    flags = [k for k, v in ctx.flags().items() if v]
    b8d = 12 * A.nnz + 4 * (A.rows + 1) + 16 * A.rows
    fmt = ctx.spmv_matrix_bytes() + 16 * A.rows
    H.product_kernels()
    ms = min(ctx.spmv_bench(100) for _ in range(3))
    kern = H.product_kernels()
    out = dict(numbering=kind, rows=A.rows, nnz=A.nnz, generate_s=round(t_gen, 1), create_s=round(t_plan, 2), flags=flags, kernels=kern,
               spmv_back_to_back_us=round(1e3 * ms, 2), survey_8d_bytes=b8d, format_bytes=fmt,
               survey_8d_gbps=round(b8d / ms / 1e6), format_gbps=round(fmt / ms / 1e6),
               frac_of_copy_8d=round(b8d / ms / 1e6 / stream["copy"], 3), frac_8tb_format=round(fmt / ms / 1e6 / 8000, 3))
    ones = np.ones(A.rows)
    bvec = ctx.spmv(ones)
    for m in a.methods.split(","):
        ctx.load(np.zeros(A.rows), bvec)
        ctx.run_begin(m, tol=0.0, max_iter=20 + a.steps, check_every=a.steps, krr=50, nrr=2)
        ctx.run_iterate(20); ctx.sync()
        t0 = time.perf_counter(); ctx.run_iterate(a.steps); ctx.sync(); dt = time.perf_counter() - t0
        res = ctx.run_end()
        out[m + "_ms_per_iteration"] = round(1e3 * dt / a.steps, 4)
        out[m + "_iterations"] = int(res.iterations)
    if not a.no_oracle:
        import oracle_lib as O
        row, col, val = A.to_coo()
        x = 1.0 + 1e-3 * np.cos(np.arange(A.rows))
        out["spmv_bitexact_vs_oracle"] = bool(np.array_equal(ctx.spmv(x), O.spmv(A.rows, row, col, val, x)))
    ctx.close()
    print(json.dumps(out), flush=True)
