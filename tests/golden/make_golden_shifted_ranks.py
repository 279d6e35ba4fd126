#!/usr/bin/env python3
"""Generate tests/golden/ranks_*.npz: the REAL reference's shifted solvers at P = 2 and 4 MPI ranks
(oracle/_ref/ref_dump_shifted / ref_dump_switching = ref_dump_shifted_main.c on top of the
reference's shifted_solver.c / shifted_switching_solver.c, run under mpiexec). The single-rank pins
live in shifted_*.npz / switching_*.npz; these fixtures pin the oracle's P-virtual-rank
restatement (distributed SpMV, dot products associated like MPICH's all-reduce) for the same
functions. Set-up as reference src/test_shifted.c:95-117: b = A*1 + sigma[seed]*1, x0 = 0."""
import os
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

FUNCS = [("ref_dump_shifted", "shifted_lopbicgstab"), ("ref_dump_shifted", "shifted_pipe_lopbicgstab"),
         ("ref_dump_shifted", "shifted_bicgstab"), ("ref_dump_switching", "shifted_lopbicg"),
         ("ref_dump_switching", "shifted_lopbicg_switching")]


def run(binary, mtx, fn, seed, sigma, P, n):
    with tempfile.TemporaryDirectory() as td:
        prefix = os.path.join(td, "o")
        cmd = [R.MPIEXEC, "-n", str(P), os.path.join(R.REF_DIR, binary), mtx, fn, prefix, str(seed)] + [repr(float(s)) for s in sigma]
        subprocess.run(cmd, check=True, capture_output=True, timeout=600)
        counts, displs = synth.partition(n, P)
        nsig = len(sigma)
        b, x, r, k = np.zeros(n), np.zeros((nsig, n)), np.zeros(n), None
        for p in range(P):
            raw = open(f"{prefix}.rank{p}.bin", "rb").read()
            kk, nl, ns = np.frombuffer(raw[:12], dtype=np.int32)
            assert nl == counts[p] and ns == nsig
            k = int(kk) if k is None else k
            assert k == int(kk)
            body = np.frombuffer(raw[12:], dtype=np.float64)
            lo = int(displs[p])
            b[lo:lo + nl] = body[:nl]
            x[:, lo:lo + nl] = body[nl:nl + nsig * nl].reshape(nsig, nl)
            r[lo:lo + nl] = body[nl + nsig * nl:]
        return k, b, x, r


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    cases = [("ranks_stencil7_m12_lin8_seed7", synth.stencil7(12), np.linspace(0.0, 3.0, 8), 7),
             ("ranks_offsets_n3001_s6_seed2", synth.from_offsets(3001, (0, 1, -1, 40, -40, 41, -41, 900, -900), diag_base=10.0, seed=6),
              0.01 * (np.arange(6) + 1.0), 2)]
    for name, A, sigma, seed in cases:
        with tempfile.TemporaryDirectory() as td:
            mtx = os.path.join(td, "a.mtx")
            synth.write_mtx(mtx, A)
            out = dict(n=A.rows, ptr=A.ptr, col=A.col, val=A.val, sigma=sigma, seed=seed)
            for P in (2, 4):
                for binary, fn in FUNCS:
                    k, b, x, r = run(binary, mtx, fn, seed, sigma, P, A.rows)
                    out[f"{fn}_P{P}_k"] = k
                    out[f"{fn}_P{P}_b"] = b
                    out[f"{fn}_P{P}_x"] = x
                    out[f"{fn}_P{P}_r"] = r
                    print(name, fn, "P =", P, "k =", k)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    main()
