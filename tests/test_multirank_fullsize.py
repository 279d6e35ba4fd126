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


def shifted_outputs(out, A, coo, world):
    """config 5 across ranks: 16 shifts sigma_j = (j+1) 0.01/16, seed 7 (bench.py's leg; reference src/main_shifted.c:99 pattern)"""
    row, col, val = coo
    nsh, seed = 16, 7
    sigma = (np.arange(nsh) + 1.0) * 0.01 / nsh
    out["shifted_methods"] = np.array(["shifted_lopbicgstab", "shifted_pipe_lopbicgstab"])
    out["shifted_sigma"], out["shifted_seed"], out["shifted_sel"] = sigma, seed, np.array([0, 7, 15])
    bs = out["b"] + sigma[seed] * np.ones(A.rows)
    for which in ("shifted_lopbicgstab", "shifted_pipe_lopbicgstab"):
        orc = O.solve_shifted(A.rows, row, col, val, bs, sigma, seed, nranks=world, tol=0.0, max_iter=K_FIX, which=which)
        assert orc["k"] == K_FIX
        for key in ("alpha", "omega", "beta", "dotr"):
            out[f"{which}_{key}"] = orc[key]
        for j in (0, 7, 15):
            out[f"{which}_x{j}"] = orc["x"][j]


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
    if world == 8:
        # plain and CA-BiCGStab all the way to 1e-9 (the pipelined recurrences stagnate above that on this matrix, SURVEY
        # section 4). The yardstick is the reference's OWN spread over rank counts (tests/golden/transport_convergence.json,
        # written by make_transport_convergence.py through the pinned oracle: plain 526 / 621 / 669 / 567 iterations at
        # P = 1 / 2 / 4 / 8, CA 596 / 617 / 563 / 540 -- the association of the dot sums alone moves the count by +-13 %)
        import json
        gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "transport_convergence.json")))
        assert gold["rows"] == A.rows and gold["nnz"] == A.nnz
        out["converge_methods"] = np.array(["bicgstab", "ca_bicgstab"])
        out["converge_tol"] = gold["tol"]
        for method in ("bicgstab", "ca_bicgstab"):
            ks = [gold["runs"][f"{method}_P{P}"]["k"] for P in (1, 2, 4, 8)]
            out[f"{method}_conv_kmin"], out[f"{method}_conv_kmax"] = min(ks), max(ks)
            out[f"{method}_conv_err"] = max(gold["runs"][f"{method}_P{P}"]["max_err_vs_ones"] for P in (1, 2, 4, 8))
        shifted_outputs(out, A, (row, col, val), world)
    _oracle_cache[world] = out
    return out


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("kind", ["host", "host-p2p"])
def test_fullsize_partition_against_oracle(matrix, world, kind):
    with tempfile.TemporaryDirectory() as td:
        out = dict(oracle_outputs(matrix, world))
        if kind != "host-p2p":        # the runs to convergence (8 ranks time-slicing one GPU: ~1 minute) and the 16-shift leg once, on the production data path
            out = {k: v for k, v in out.items() if "conv" not in k and "shifted" not in k}
        np.savez(os.path.join(td, "oracle.npz"), **out)
        mp.start_processes(W.fullsize_worker, args=(world, _free_port(), kind, td), nprocs=world, join=True,
                           start_method="spawn")
        fails = glob.glob(os.path.join(td, "fail*"))
        assert not fails, open(fails[0]).read()
        assert len(glob.glob(os.path.join(td, "ok*"))) == world


@pytest.mark.parametrize("numbering,world,kind", [("rcm", 8, "host-p2p"), ("rcm", 8, "host"), ("random", 8, "host-p2p"), ("generator", 8, "host"),
                                                  ("rcm", 2, "host-p2p"), ("generator", 2, "host-p2p")])
def test_unstructured_mesh_partition_against_oracle(numbering, world, kind, tmp_path_factory):
    """The unstructured FEM matrix (mpi_bicgstab_amd.mesh: 1 601 613 ragged rows, the stand-in for Transport.mtx) in the reference's
    row partition (src/matrix.c:295-308) across ranks sharing the GPU: RCM numbering at 8 ranks through both transports (halo = the
    neighbouring level sets), the random permutation likewise (every rank needs nearly all of x: the halo IS the vector); at 2 ranks
    (800 k rows each) the exchange runs as separate launches and the halo-free rows go through k_spmv_jagd / k_spmv_jagw (asserted). Distributed
    SpMV bit-exact against the oracle at the same P (src/matrix.c:428-441), first 8 iterations of the four solvers against its
    alpha / omega / beta / (r,r)."""
    from mpi_bicgstab_amd import mesh
    cache = tmp_path_factory.getbasetemp() / "mesh_cache"
    cache.mkdir(exist_ok=True)
    A = mesh.fem_unstructured(117, numbering, scale_decades=SCALE_DECADES, cache_dir=str(cache))
    row, col, val = A.to_coo()
    # (8 iterations at rtol 1e-6: the reference's own scalars spread by a factor of ten per iteration on this matrix when only the
    # association of the dot sums changes, tests/test_mesh_gpu.py)
    k_fix = 8
    out = dict(n=A.rows, k_fix=k_fix, rtol=1e-6, scale_decades=SCALE_DECADES, mesh_numbering=numbering, mesh_m=117, mesh_cache=str(cache))
    out["x_in"] = np.random.default_rng(99).standard_normal(A.rows)
    out["y"] = O.spmv(A.rows, row, col, val, out["x_in"], nranks=world)
    out["b"] = O.spmv(A.rows, row, col, val, np.ones(A.rows), nranks=world)
    for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"):
        orc = O.solve(method, A.rows, row, col, val, out["b"], nranks=world, tol=0.0, max_iter=k_fix, krr=5, nrr=1)
        assert orc["k"] == k_fix
        for key in ("alpha", "omega", "beta", "dotr", "x"):
            out[f"{method}_{key}"] = orc[key]
    del A, row, col, val
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "oracle.npz"), **out)
        mp.start_processes(W.fullsize_worker, args=(world, _free_port(), kind, td), nprocs=world, join=True, start_method="spawn")
        fails = glob.glob(os.path.join(td, "fail*"))
        assert not fails, open(fails[0]).read()
        assert len(glob.glob(os.path.join(td, "ok*"))) == world


@pytest.mark.parametrize("wide", [0, 2])
def test_laplace7_slabs_8_ranks_ca_bicgstab(wide):
    """BASELINE.json configs[3] (7-point Laplacian 512^3 over 8 GPUs = 64 planes of 512^2 each, CA-BiCGStab) at 128^3:
    8 ranks x 16-plane z-slabs in the reference's row partition, peer-to-peer data path (halo = one plane per neighbour,
    exchanged inside the SpMV launch): distributed SpMV bit-exact, first 12 iterations of ca_bicgstab and bicgstab against
    the oracle at 8 ranks (src/solver.c:160-278). wide = 2: the halo-free planes through the two-rows-per-lane form of the
    plane-marching product (what a 64-plane slab of 512^2 takes by default)."""
    m, world = 128, 8
    A = synth.stencil7(m, synth.LAPLACE_WEIGHTS)
    row, col, val = A.to_coo()
    out = dict(n=A.rows, k_fix=K_FIX, scale_decades=0.0, grid=m, methods=np.array(["ca_bicgstab", "bicgstab"]))
    if wide:
        out["stencil_wide"] = wide
    out["x_in"] = np.random.default_rng(77).standard_normal(A.rows)
    out["y"] = O.spmv(A.rows, row, col, val, out["x_in"], nranks=world)
    out["b"] = O.spmv(A.rows, row, col, val, np.ones(A.rows), nranks=world)
    for method in ("ca_bicgstab", "bicgstab"):
        orc = O.solve(method, A.rows, row, col, val, out["b"], nranks=world, tol=0.0, max_iter=K_FIX)
        for key in ("alpha", "omega", "beta", "dotr", "x"):
            out[f"{method}_{key}"] = orc[key]
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "oracle.npz"), **out)
        mp.start_processes(W.fullsize_worker, args=(world, _free_port(), "host-p2p", td), nprocs=world, join=True, start_method="spawn")
        fails = glob.glob(os.path.join(td, "fail*"))
        assert not fails, open(fails[0]).read()
        assert len(glob.glob(os.path.join(td, "ok*"))) == world


@pytest.mark.parametrize("share", [8, 4])
def test_two_small_ranks_persistent_with_halo(share):
    """The form an 8-GPU run of BASELINE.json configs[2] takes -- ONE persistent launch per chunk of iterations with the
    neighbours' halo values arriving as LL words in the landing ring and the dot sums crossing the mailboxes -- at a rank size
    the one-GPU box can hold TWICE: two processes x 100 132 rows of a 200 264-row Transport-shaped matrix (each fits the 127
    workgroups a rank gets when two share the 256 CUs; the 8-rank full-size tests above get 31 and fall back to the two-launch
    form). `persist` AND halo > 0 asserted on both ranks; distributed SpMV bit-exact, the first 12 iterations of all four
    solvers and of the two 16-shift solvers against the oracle at P = 2 (reference src/matrix.c:428-441, src/solver.c:351-398,
    src/shifted_solver.c:257-319). share = 4: two processes x 200 264 rows -- 1 577 rows per CU of a rank's 127, the rank size of
    the headline at 4 GPUs: TWO rows per thread in the persistent kernels of all three methods (round 6: plain and CA-BiCGStab
    too), asserted by the absence of product kernels during their iterations; the shifted solvers (one row per thread only) keep
    their launches there."""
    world = 2
    n = (synth.TRANSPORT_N + share - 1) // share
    A = synth.transport_like(n=n, scale_decades=SCALE_DECADES)
    row, col, val = A.to_coo()
    out = dict(n=n, k_fix=K_FIX, scale_decades=SCALE_DECADES, expect_persist=1)
    if share == 4:
        out["expect_no_product_kernels"] = np.array(["bicgstab", "ca_bicgstab", "pipe_bicgstab"])
    out["x_in"] = np.random.default_rng(31).standard_normal(n)
    out["y"] = O.spmv(n, row, col, val, out["x_in"], nranks=world)
    out["b"] = O.spmv(n, row, col, val, np.ones(n), nranks=world)
    for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"):
        orc = O.solve(method, n, row, col, val, out["b"], nranks=world, tol=0.0, max_iter=K_FIX, krr=5, nrr=1)
        for key in ("alpha", "omega", "beta", "dotr", "x"):
            out[f"{method}_{key}"] = orc[key]
    shifted_outputs(out, A, (row, col, val), world)
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "oracle.npz"), **out)
        mp.start_processes(W.fullsize_worker, args=(world, _free_port(), "host-p2p", td), nprocs=world, join=True, start_method="spawn")
        fails = glob.glob(os.path.join(td, "fail*"))
        assert not fails, open(fails[0]).read()
        assert len(glob.glob(os.path.join(td, "ok*"))) == world
