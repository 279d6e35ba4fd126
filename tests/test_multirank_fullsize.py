"""N > 1 parity at BASELINE.json's full size (`-m gpu`): 4 and 8 ranks sharing the one GPU of the box
hold the reference's row partition (src/matrix.c:295-308) of the Transport-shaped matrix (1 602 111
rows; 200 k-row slabs with +-13 807-column halos at 8 ranks = BASELINE.json configs[2]), through
both transports (gloo-staged collectives, and the peer-to-peer data path with the halo exchange
folded into the SpMV launch). Checks: distributed SpMV bit-exact against the oracle at the same P
(reference src/matrix.c:428-441), first 12 iterations of all four solvers against the oracle's
alpha/omega/beta/(r,r) trajectory (src/solver.c:363-385 for the pipelined overlap)."""
import glob
import os
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

import mp_workers as W
import oracle_lib as O
from mpi_bicgstab_amd import synth
from test_multirank import _free_port

pytestmark = pytest.mark.gpu

K_FIX = 12
SCALE_DECADES = 2.0


@pytest.fixture(scope="module")
def matrix():
    A = synth.transport_like(scale_decades=SCALE_DECADES)
    return A, A.to_coo()


_oracle_cache = {}


def oracle_outputs(matrix, world):
    """one CPU run of the oracle per rank count (its dot products associate per rank, src/solver.c:89-91)"""
    if world in _oracle_cache:
        return _oracle_cache[world]
    A, (row, col, val) = matrix
    out = dict(n=A.rows, k_fix=K_FIX, scale_decades=SCALE_DECADES)
    out["x_in"] = np.random.default_rng(2024).standard_normal(A.rows)
    out["y"] = O.spmv(A.rows, row, col, val, out["x_in"], nranks=world)
    out["b"] = O.spmv(A.rows, row, col, val, np.ones(A.rows), nranks=world)
    for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"):
        orc = O.solve(method, A.rows, row, col, val, out["b"], nranks=world, tol=0.0, max_iter=K_FIX, krr=5, nrr=1)
        assert orc["k"] == K_FIX
        for key in ("alpha", "omega", "beta", "dotr", "x"):
            out[f"{method}_{key}"] = orc[key]
    _oracle_cache[world] = out
    return out


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("kind", ["host", "host-p2p"])
def test_fullsize_partition_against_oracle(matrix, world, kind):
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "oracle.npz"), **oracle_outputs(matrix, world))
        mp.start_processes(W.fullsize_worker, args=(world, _free_port(), kind, td), nprocs=world, join=True,
                           start_method="spawn")
        fails = glob.glob(os.path.join(td, "fail*"))
        assert not fails, open(fails[0]).read()
        assert len(glob.glob(os.path.join(td, "ok*"))) == world
