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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=synth.TRANSPORT_N)
    ap.add_argument("--scale-decades", type=float, default=2.0)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--method", default="bicgstab")
    a = ap.parse_args()

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
