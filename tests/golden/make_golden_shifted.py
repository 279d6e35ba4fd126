#!/usr/bin/env python3
"""Generate tests/golden/shifted_*.npz from the REAL reference's shifted_lopbicgstab
(oracle/_ref/libref_shifted.so = reference src/shifted_solver.c compiled unmodified; EPS 1e-12,
MAX_ITER 1000). Set-up follows reference src/test_shifted.c:95-111: sigma_j = 0.01 (j+1),
b = (A + sigma_seed I) * 1, x0 = 0. Also stores whether the _v2 / _nooverlap variants are
bit-identical to the base function (they are)."""
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


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    cases = [("shifted_stencil7_m10_s5_seed0", synth.stencil7(10), 5, 0),
             ("shifted_stencil7_m10_s5_seed2", synth.stencil7(10), 5, 2),
             ("shifted_offsets_n3001_s16_seed7", synth.from_offsets(3001, (0, 1, -1, 40, -40, 41, -41, 900, -900), diag_base=10.0, seed=6), 16, 7)]
    for name, A, nsig, seed in cases:
        M = R.RefMatrix(A)
        sigma = 0.01 * (np.arange(nsig) + 1.0) if nsig == 5 else (np.arange(nsig) + 1.0) * 0.01 / 16     # src/test_shifted.c:97, main_shifted.c:99 pattern
        b = R.spmv(M, np.ones(A.rows))
        b = b + sigma[seed] * np.ones(A.rows)                 # my_daxpy(sigma[seed], 1, b), src/test_shifted.c:111
        ref = R.solve_shifted("shifted_lopbicgstab", M, b, sigma, seed)
        same = all(np.array_equal(R.solve_shifted(f, M, b, sigma, seed)["x"], ref["x"])
                   for f in ("shifted_lopbicgstab_v2", "shifted_lopbicgstab_nooverlap"))
        pipe = R.solve_shifted("shifted_pipe_lopbicgstab", M, b, sigma, seed)
        pipe_same = np.array_equal(R.solve_shifted("shifted_pipe_lopbicgstab_nooverlap", M, b, sigma, seed)["x"], pipe["x"])
        b0 = R.spmv(M, np.ones(A.rows))                        # shifted_bicgstab: the seed system is A itself
        xi = R.solve_shifted("shifted_bicgstab", M, b0, sigma, 0)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), n=A.rows, ptr=A.ptr, col=A.col, val=A.val, sigma=sigma,
                            seed=seed, b=b, k=ref["k"], x=ref["x"], r=ref["r"], variants_bit_identical=same,
                            pipe_k=pipe["k"], pipe_x=pipe["x"], pipe_r=pipe["r"], pipe_variants_bit_identical=pipe_same,
                            xi_b=b0, xi_k=xi["k"], xi_x=xi["x"], xi_r=xi["r"])
        print(name, "k =", ref["k"], pipe["k"], xi["k"], "variants identical:", same, pipe_same)


if __name__ == "__main__":
    main()
