#!/usr/bin/env python3
"""CPU baseline leg of bench.py (runs as a subprocess, prints ONE JSON object).

Times the reference's own bicgstab() -- oracle/_ref/libref_env.so = reference solver.c / matrix.c /
vector.c compiled at -O3 -march=x86-64-v3 from /root/reference by oracle/Makefile, with MAX_ITER
and EPS made overridable (kind "reference") -- on ONE host core (one MPI rank, MPICH singleton
init), on the same synthetic workload as the GPU leg, for a bounded number of iterations.
Falls back to the plain-C restatement oracle/liboracle.so (kind "port") when the reference build
is not present. The value is the reference's own "Avg time/iter" definition: wall time from
before the set-up SpMV to the last iteration, divided by k (src/solver.c:70,130-139).
"""
from __future__ import annotations

import argparse
import contextlib
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from mpi_bicgstab_amd import synth  # noqa: E402


@contextlib.contextmanager
def quiet_stdout():
    """the reference prints its summary with printf; keep our stdout = one JSON line"""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)


def rank_worker(a):
    """One MPI rank of the reference run on `--ranks` host cores (started by mpiexec): this rank's row
    slab of the same matrix, b = A*1 through the reference's own MPI_csr_spmv_ovlap, then the
    reference's bicgstab() collectively over MPI_COMM_WORLD. Rank 0 prints the JSON object."""
    import ref_lib as R
    libref = os.path.join(ROOT, "oracle", "_ref", "libref_env.so")
    R.LIBREF = libref
    os.environ["REF_MAX_ITER"] = str(a.iters)
    os.environ["REF_EPS"] = "0"
    with quiet_stdout():
        lib = R.lib()                                   # MPI_Init under mpiexec: joins MPI_COMM_WORLD
    mpi = C.CDLL(R.MPI_LIB, mode=C.RTLD_GLOBAL)
    world = C.c_int(0x44000000)                         # MPICH's MPI_COMM_WORLD handle
    rank, size = C.c_int(0), C.c_int(1)
    mpi.MPI_Comm_rank(world, C.byref(rank)); mpi.MPI_Comm_size(world, C.byref(size))
    rank, size = rank.value, size.value
    counts, displs = synth.partition(a.n, size)
    lo, hi = int(displs[rank]), int(displs[rank] + counts[rank])
    slab = synth.transport_like(n=a.n, rows=(lo, hi), scale_decades=a.scale_decades)
    diag, offd = synth.split_row_slab(slab, lo)
    keep = []

    def csr(M, ncols):
        v = np.ascontiguousarray(M.val, dtype=np.float64); c = np.ascontiguousarray(M.col, dtype=np.uint32)
        p = np.ascontiguousarray(M.ptr, dtype=np.uint32)
        keep.extend([v, c, p])
        dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint)
        return R.CSRMatrix(v.ctypes.data_as(dp), c.ctypes.data_as(up), p.ctypes.data_as(up), int(p[-1]), M.rows, ncols)
    d, o = csr(diag, hi - lo), csr(offd, a.n)
    cnt = np.ascontiguousarray(counts, dtype=np.int32); dsp = np.ascontiguousarray(displs, dtype=np.int32)
    ip = C.POINTER(C.c_int)
    info = R.InfoMatrix(synth.transport_nnz(a.n), a.n, a.n, b"MCRG", cnt.ctypes.data_as(ip), dsp.ctypes.data_as(ip))
    dp = C.POINTER(C.c_double)
    ones, full, b = np.ones(hi - lo), np.zeros(a.n), np.zeros(hi - lo)
    lib.MPI_csr_spmv_ovlap(C.byref(d), C.byref(o), C.byref(info), ones.ctypes.data_as(dp), full.ctypes.data_as(dp), b.ctypes.data_as(dp))
    x = np.zeros(hi - lo)
    fn = getattr(lib, a.method)
    fn.restype = C.c_int
    mpi.MPI_Barrier(world)
    with quiet_stdout():
        t0 = time.perf_counter()
        k = fn(C.byref(d), C.byref(o), C.byref(info), x.ctypes.data_as(dp), b.ctypes.data_as(dp))
        dt = time.perf_counter() - t0
    mpi.MPI_Barrier(world)
    if rank == 0:
        print(json.dumps(dict(unit="ms/iteration", cores=size, kind="reference", value=1e3 * dt / max(k, 1), iterations=int(k),
                              sample=f"{a.iters} iterations of {a.method} on the full workload (n={a.n}), {size} MPI ranks "
                                     "(reference row partition, MPI_Iallgatherv + MPI_Iallreduce over shared memory)",
                              flags="clang -O3 -march=x86-64-v3 (reference Makefile: mpifccpx -Kfast)")), flush=True)
    mpi.MPI_Finalize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=synth.TRANSPORT_N)
    ap.add_argument("--scale-decades", type=float, default=2.0)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--method", default="bicgstab")
    ap.add_argument("--ranks", type=int, default=1, help="> 1: the reference on that many host cores (mpiexec)")
    ap.add_argument("--rank-worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.rank_worker:
        return rank_worker(a)
    if a.ranks > 1:
        import subprocess
        cmd = ["/opt/conda/bin/mpiexec", "-n", str(a.ranks), sys.executable, os.path.abspath(__file__), "--rank-worker",
               "--n", str(a.n), "--scale-decades", str(a.scale_decades), "--iters", str(a.iters), "--method", a.method]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not lines:
            print(json.dumps({"error": (out.stderr or out.stdout)[-400:]}))
            return
        print(lines[-1])
        return

    A = synth.transport_like(n=a.n, scale_decades=a.scale_decades)
    b = A.matvec(np.ones(A.rows))
    out = dict(unit="ms/iteration", cores=1,
               sample=f"{a.iters} iterations of {a.method} on the full workload (n={A.rows}, nnz={A.nnz}), 1 MPI rank")
    libref = os.path.join(ROOT, "oracle", "_ref", "libref_env.so")
    if os.path.exists(libref) and os.path.exists("/opt/conda/lib/libmpi.so"):
        os.environ["REF_MAX_ITER"] = str(a.iters)
        os.environ["REF_EPS"] = "0"
        import ref_lib as R
        R.LIBREF = libref
        M = R.RefMatrix(A)
        with quiet_stdout():
            R.lib()
            t0 = time.perf_counter()
            res = R.solve(a.method, M, b)
            dt = time.perf_counter() - t0
        out.update(kind="reference", value=1e3 * dt / max(res["k"], 1), iterations=int(res["k"]),
                   flags="clang -O3 -march=x86-64-v3 (reference Makefile: mpifccpx -Kfast)")
    else:
        import oracle_lib as O
        row, col, val = A.to_coo()
        t0 = time.perf_counter()
        res = O.solve(a.method, A.rows, row, col, val, b, tol=0.0, max_iter=a.iters)
        dt = time.perf_counter() - t0
        out.update(kind="port", value=1e3 * dt / max(res["k"], 1), iterations=int(res["k"]),
                   flags="gcc -O2 -ffp-contract=off (includes the COO->CSR build)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
