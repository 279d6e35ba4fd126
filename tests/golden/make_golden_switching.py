#!/usr/bin/env python3
"""Generate tests/golden/switching_*.npz from the REAL reference's seed-switching shifted solvers
(oracle/_ref/libref_switching.so = reference src/shifted_switching_solver.c compiled unmodified;
EPS 1e-12, MAX_ITER 1000): shifted_lopbicg (:20-257), shifted_lopbicg_switching (:260-608) and
whether shifted_lopbicg_switching_noovlp (:611-1016) is bit-identical to it (it is).
Set-up as reference src/main_shifted.c:95-117: b = (A + sigma_seed I) * 1, x0 = 0. The shift sets are
chosen so that the seed converges first and a seed switch really happens in all but the first case."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mpi_bicgstab_amd import synth  # noqa: E402
import ref_lib as R  # noqa: E402


def cases():
    off = synth.from_offsets(3001, (0, 1, -1, 40, -40, 41, -41, 900, -900), diag_base=10.0, seed=6)
    return [("switching_stencil7_m10_s5_seed2", synth.stencil7(10), 0.01 * (np.arange(5) + 1.0), 2),
            ("switching_stencil7_m12_lin8_seed7", synth.stencil7(12), np.linspace(0.0, 3.0, 8), 7),
            ("switching_offsets_n3001_geo12_seed11", off, 0.01 * 2.0 ** np.arange(12), 11),
            ("switching_transport_n6000_geo10_seed9", synth.transport_like(n=6000, scale_decades=1.0), 0.02 * 2.0 ** np.arange(10), 9)]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name, A, sigma, seed in cases():
        M = R.RefMatrix(A)
        b = R.spmv(M, np.ones(A.rows)) + sigma[seed] * np.ones(A.rows)
        flag = R.solve_switching("shifted_lopbicg", M, b, sigma, seed)
        sw = R.solve_switching("shifted_lopbicg_switching", M, b, sigma, seed)
        no = R.solve_switching("shifted_lopbicg_switching_noovlp", M, b, sigma, seed)
        same = no["k"] == sw["k"] and np.array_equal(no["x"], sw["x"]) and np.array_equal(no["r"], sw["r"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), n=A.rows, ptr=A.ptr, col=A.col, val=A.val, sigma=sigma, seed=seed,
                            b=b, flag_k=flag["k"], flag_x=flag["x"], flag_r=flag["r"], sw_k=sw["k"], sw_x=sw["x"], sw_r=sw["r"],
                            noovlp_bit_identical=same)
        print(name, "k =", flag["k"], sw["k"], "noovlp identical:", same)


if __name__ == "__main__":
    main()
