"""Iteration counts of the reference to tol 1e-9 on the Transport-shaped synthetic (bench.py's headline matrix, 1 602 111 rows,
values scaled over 2 decades) at P = 1, 2, 4, 8 ranks -- through oracle/liboracle.so, which tests/test_oracle_golden.py pins
bit for bit to the real reference (oracle/_ref/libref.so, built from /root/reference/src). ~1 minute per solve on one core, so
the numbers are committed (transport_convergence.json) instead of being recomputed by every GPU test run.

    python tests/golden/make_transport_convergence.py

What they show: the reference's own count moves by +-13 % with the rank count alone (the association of its dot products,
src/solver.c:89-91) -- the spread a GPU run has to fall into; there is no single "right" count at this tolerance."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from mpi_bicgstab_amd import synth  # noqa: E402

TOL = 1e-9
A = synth.transport_like(scale_decades=2.0)
row, col, val = A.to_coo()
out = {"matrix": "synth.transport_like(scale_decades=2.0)", "rows": A.rows, "nnz": A.nnz, "tol": TOL, "runs": {}}
for P in (1, 2, 4, 8):
    b = O.spmv(A.rows, row, col, val, np.ones(A.rows), nranks=P)
    for method in ("bicgstab", "ca_bicgstab"):
        o = O.solve(method, A.rows, row, col, val, b, nranks=P, tol=TOL, max_iter=4000)
        out["runs"][f"{method}_P{P}"] = {"k": int(o["k"]), "max_err_vs_ones": float(np.abs(o["x"] - 1.0).max()),
                                         "relres": float(np.sqrt(o["dot_r"] / o["dot_zero"]))}
        print(P, method, out["runs"][f"{method}_P{P}"], flush=True)
json.dump(out, open(os.path.join(HERE, "transport_convergence.json"), "w"), indent=1)
