#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref, built from /root/reference).

Run in the build container only (needs /root/reference for `make -C oracle ref`, and MPICH):

    python tests/golden/make_golden.py

Each fixture holds a small matrix (CSR, ascending columns = the stored order the reference gets
from a column-major .mtx file after its stable row sort, reference src/matrix.c:135-183,206-232),
b = A*1 computed BY THE REFERENCE at each rank count (b_P1, b_P2, b_P4: the diag-then-offd
summation of src/matrix.c:437-440 makes b depend on P in the last bits; src/main.c:109-117), and for every solver x P combination the
reference's outputs: iteration count k, solution x, recursive residual r (concatenated rank
blocks). P = 1 goes through libref.so in-process; P = 2, 4, 8 run oracle/_ref/ref_dump under mpiexec
on a .mtx file written with 17 significant digits.

The reference constants are compiled in: EPS = 1e-15, MAX_ITER = 1000 (src/solver.c:3-4).
"""
from __future__ import annotations

import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mpi_bicgstab_amd import synth  # noqa: E402
import ref_lib as R  # noqa: E402

SOLVERS = [("bicgstab", ()), ("ca_bicgstab", ()), ("pipe_bicgstab", ()), ("pipe_bicgstab_rr", (10, 3))]


def cases():
    yield "stencil7_m8", synth.stencil7(8)                       # 512 rows, nonsymmetric weights
    yield "stencil7_m12", synth.stencil7(12)                     # the SURVEY.md section 4 probe
    yield "band_n700_b5", synth.banded(700, 5, diag_base=9.0)
    yield "offsets_n1501", synth.from_offsets(1501, (0, 1, -1, 7, -7, 8, -8, 113, -113, 120, -120),
                                              diag_base=6.0, seed=99)
    yield "ragged_n400", synth.random_rows(400, 12, seed=5, empty_frac=0.0, long_rows={17: 300})


def run_ref_dump(mtx, method, extra, nranks, n):
    with tempfile.TemporaryDirectory() as td:
        prefix = os.path.join(td, "out")
        cmd = [R.MPIEXEC, "-n", str(nranks), os.path.join(R.REF_DIR, "ref_dump"), mtx, method, prefix]
        cmd += [str(e) for e in extra]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        xs, rs, k = [], [], None
        for p in range(nranks):
            raw = open(f"{prefix}.rank{p}.bin", "rb").read()
            kk, nl = struct.unpack("ii", raw[:8])
            k = kk
            body = np.frombuffer(raw[8:], dtype=np.float64)
            xs.append(body[:nl])
            rs.append(body[nl:2 * nl])
        x, r = np.concatenate(xs), np.concatenate(rs)
        assert len(x) == n
        return k, x, r


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name, A in cases():
        M = R.RefMatrix(A)
        out = dict(n=A.rows, ptr=A.ptr, col=A.col, val=A.val)
        out["b"] = out["b_P1"] = R.spmv(M, np.ones(A.rows))
        xprobe = 1.0 + 0.001 * np.arange(A.rows)
        out["spmv_x"] = xprobe
        out["spmv_y_P1"] = R.spmv(M, xprobe)
        out["dot_b_b"] = R.ddot(out["b"], out["b"])
        with tempfile.TemporaryDirectory() as td:
            mtx = os.path.join(td, name + ".mtx")
            synth.write_mtx(mtx, A)
            for P in (2, 4, 8):
                _, y, _ = run_ref_dump(mtx, "spmv", (), P, A.rows)
                out[f"spmv_y_P{P}"] = y
                _, _, bP = run_ref_dump(mtx, "rhs", (), P, A.rows)
                out[f"b_P{P}"] = bP
            for method, extra in SOLVERS:
                res = R.solve(method, M, out["b"], *(extra or (0, 0)))
                out[f"{method}_P1_k"], out[f"{method}_P1_x"], out[f"{method}_P1_r"] = res["k"], res["x"], res["r"]
                for P in (2, 4, 8):
                    k, x, r = run_ref_dump(mtx, method, extra, P, A.rows)
                    out[f"{method}_P{P}_k"], out[f"{method}_P{P}_x"], out[f"{method}_P{P}_r"] = k, x, r
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "n =", A.rows, "nnz =", A.nnz,
              {m: [int(out[f"{m}_P{P}_k"]) for P in (1, 2, 4, 8)] for m, _ in SOLVERS})


if __name__ == "__main__":
    main()
