"""Parity of every workload bench.py TIMES, at the size it times it (`-m gpu`).

bench.py's headline matrix is covered by tests/test_full_size.py; this module builds the other legs with the very
generator calls of bench.py's build() -- synthetic banded CSR at half-bandwidth 8 / 64 / 512 (~24 M non-zeros each),
the FEM-like irregular matrix on 1.6 M rows, the 256^3 Laplacian (one GPU's share of BASELINE.json configs[3]) and
configs[4] exactly as benchmarked (16 shifts, seed 7, Transport-shaped) -- and compares the HIP path with the oracle:

  * SpMV against mult() (reference src/matrix.c:498-516): bit for bit where a lane adds its row in stored order;
    1e-13 x sum_j |a_ij x_j| where a row is spread over several lanes (long rows, BICG_FLAG_ROWSPLIT) -- stated
    tolerance, north_star's bar for floating point;
  * the first 12 iterations of the methods bench.py times on that matrix against the oracle's alpha / omega / beta /
    (r,r) trajectory at rtol 1e-7, and the iterate itself;
  * the shifted solvers: first 12 seed scalars, every x_j, and the per-shift residuals of the reference's check
    (src/test_shifted.c:129-154) through bicg_shifted_residuals.
"""
import numpy as np
import pytest

import oracle_lib as O
from mpi_bicgstab_amd import hipsolver as H
from mpi_bicgstab_amd import synth

pytestmark = pytest.mark.gpu

SCALE = 2.0          # bench.py --scale-decades default
K = 12


def _spmv_check(ctx, A, coo, seed=0):
    row, col, val = coo
    x = np.random.default_rng(seed).standard_normal(A.rows)
    y, y_orc = ctx.spmv(x), O.spmv(A.rows, row, col, val, x)
    if ctx.flags().get("rowsplit"):
        mag = O.spmv(A.rows, row, col, np.abs(val), np.abs(x))          # sum_j |a_ij x_j| per row
        assert np.all(np.abs(y - y_orc) <= 1e-13 * mag), float(np.max(np.abs(y - y_orc) / mag))
    else:
        assert np.array_equal(y, y_orc)
    return x, y


def _trajectory_check(ctx, A, coo, methods, k=K):
    row, col, val = coo
    b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
    for method in methods:
        kw = dict(krr=5, nrr=1) if method.endswith("_rr") else {}
        orc = O.solve(method, A.rows, row, col, val, b, tol=0.0, max_iter=k, **kw)
        got = ctx.solve(method, b, tol=0.0, max_iter=k, check_every=k, **kw)
        assert got["k"] == orc["k"] == k, method
        tr = ctx.trace(k)
        for key in ("alpha", "omega", "beta", "dotr"):
            np.testing.assert_allclose(tr[key], orc[key], rtol=1e-7, err_msg=f"{method} {key}")
        assert np.abs(got["x"] - orc["x"]).max() <= 1e-8 * np.abs(orc["x"]).max(), method
        true_r = b - O.spmv(A.rows, row, col, val, got["x"])
        assert np.linalg.norm(true_r - got["r"]) <= 1e-9 * np.linalg.norm(b), method


@pytest.mark.parametrize("hb", [8, 64, 512])
def test_banded_as_benchmarked(hb):
    """extras.banded_b{8,64,512}: bench.py build("banded", 0, hb), methods bicgstab + pipe_bicgstab"""
    H.lib().bicg_comm_init_single(0)
    rows = synth.banded_rows_for(24_000_000, hb)
    A = synth.banded(rows, hb, scale_decades=SCALE)
    assert A.nnz == synth.banded_nnz(rows, hb)
    coo = A.to_coo()
    ctx = H.Context(H.single_rank_blocks(A))
    fl = ctx.flags()
    if hb == 512:
        # 1025 entries per row: lane = row would leave 92 workgroups for 256 CUs -- the rows are spread over lanes
        assert fl.get("rowsplit"), fl
    else:
        # (b = 64: 129 entries per row over 727 workgroups -- lane = row measured faster: 43.7 vs 54.2 us per SpMV)
        assert fl["all_sell"] and fl["col16"] and not fl.get("rowsplit"), fl
    _spmv_check(ctx, A, coo, seed=hb)
    _trajectory_check(ctx, A, coo, ("bicgstab", "pipe_bicgstab"))
    ctx.close()


def test_fem_like_as_benchmarked():
    """extras.fem_like: 1.6 M rows, 6..27 entries per row, jagged slices + x windows + rows dealt to lanes by length"""
    H.lib().bicg_comm_init_single(0)
    A = synth.fem_like(scale_decades=SCALE)
    coo = A.to_coo()
    ctx = H.Context(H.single_rank_blocks(A))
    fl = ctx.flags()
    assert fl["jagged"] and fl["window"] and fl["all_sell"], fl
    _spmv_check(ctx, A, coo, seed=3)
    _trajectory_check(ctx, A, coo, ("bicgstab", "pipe_bicgstab"))
    # the SpMM verification path on the layout a real FEM matrix gets (BASELINE.json configs[4] "batched SpMV")
    assert fl["spmm"], fl
    row, col, val = coo
    X = np.random.default_rng(9).standard_normal((16, A.rows))
    sigma = (np.arange(16) + 1.0) * 0.01 / 16
    Y, _ = ctx.spmm(X, sigma)
    for j in (0, 9, 15):
        assert np.array_equal(Y[j], O.spmv(A.rows, row, col, val, X[j]) + sigma[j] * X[j]), j
    # round 4: the windowed kernel on the SpMV's own x windows (slots and runs), 16 vectors for the price of a few products
    ms = min(ctx.spmm(X, sigma)[1] for _ in range(3))
    one = ctx.spmv_bench(50)
    print(f"fem_like: SpMM 16 vectors {1e3 * ms:.1f} us, one SpMV {1e3 * one:.1f} us")
    # (round 5: one product is 47 us since the ragged-rows kernel of csrc/bicg_jagw.hip and ordinary matrix loads, was 61: the
    # SpMM, unchanged at ~390 us, was 8.4 x one product. Round 6: the pipeline for ragged rows, k_spmm_jpipe, ~320 us = 7.4 x)
    assert ctx.last_spmm_kind() == "pipelined" and ms <= 8.5 * one
    ctx.close()


def test_laplace7_256_as_benchmarked():
    """extras.laplace7_256_ca: 16.8 M rows / 117 M non-zeros, methods ca_bicgstab + bicgstab"""
    H.lib().bicg_comm_init_single(0)
    A = synth.stencil7(256, synth.LAPLACE_WEIGHTS)
    assert A.nnz == synth.stencil7_nnz(256)
    coo = A.to_coo()
    ctx = H.Context(H.single_rank_blocks(A))
    assert ctx.plan_info()["sell_rows"] == A.rows
    _spmv_check(ctx, A, coo, seed=4)
    # (round 5: at this size the element-wise phases run as contiguous non-temporal tiles -- k_vec<.., TILE> -- in all three
    # reduction modes: ticket (plain, CA) and consumer-side finish (the pipelined solvers, incl. a replacement step)
    assert ctx.stencil_info()["on"] == 1
    _trajectory_check(ctx, A, coo, ("ca_bicgstab", "bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"), k=8)
    ctx.close()


@pytest.mark.parametrize("which", ["shifted_lopbicgstab", "shifted_pipe_lopbicgstab"])
def test_config5_as_benchmarked(which):
    """BASELINE.json configs[4] as bench.py runs it: Transport-shaped, 16 shifts sigma_j = (j+1) 0.01/16, seed 7,
    b = (A + sigma_seed I) 1 (reference src/main_shifted.c:99 pattern, src/test_shifted.c:95-154)"""
    H.lib().bicg_comm_init_single(0)
    A = synth.transport_like(scale_decades=SCALE)
    row, col, val = A.to_coo()
    nsh, seed = 16, 7
    sigma = (np.arange(nsh) + 1.0) * 0.01 / nsh
    ones = np.ones(A.rows)
    ctx = H.Context(H.single_rank_blocks(A))
    b = ctx.spmv(ones) + sigma[seed] * ones
    assert np.array_equal(b, O.spmv(A.rows, row, col, val, ones) + sigma[seed] * ones)
    orc = O.solve_shifted(A.rows, row, col, val, b, sigma, seed, tol=0.0, max_iter=K, which=which)
    got = ctx.solve_shifted(b, sigma, seed, tol=0.0, max_iter=K, check_every=K, which=which)
    assert got["k"] == orc["k"] == K
    tr = ctx.trace(K)
    for key in ("alpha", "omega", "beta", "dotr"):
        np.testing.assert_allclose(tr[key], orc[key], rtol=1e-7, err_msg=f"{which} {key}")
    assert np.abs(got["x"] - orc["x"]).max() <= 1e-8 * np.abs(orc["x"]).max()
    # the reference's verification loop on the device (one pass over A for the 16 shifts) against the same loop on the
    # oracle's iterates, computed with the oracle's SpMV
    rel = ctx.shifted_residuals(got["x"], b, sigma)
    nb = np.linalg.norm(b)
    want = np.array([np.linalg.norm(O.spmv(A.rows, row, col, val, orc["x"][j]) + sigma[j] * orc["x"][j] - b) / nb for j in range(nsh)])
    np.testing.assert_allclose(rel, want, rtol=1e-6)
    ctx.close()


def test_transport_rank_of_8_as_benchmarked():
    """extras.transport_rank_of_8 (BASELINE.json configs[2]: what ONE of 8 GPUs holds): bench.py build("transport", n=(N+7)//8)
    = 200 264 rows. This size takes the LDS-resident instantiation of the persistent kernels (k_pipe_persist / k_plain_persist /
    k_ca_persist<LDSMAT = true>: 241 workgroups, matrix slices + window in 155 KB of LDS) -- the `persist` flag is asserted,
    and the first 12 iterations of the three solvers that have the form against the oracle's trajectory (reference
    src/solver.c:351-398, 86-120, 216-251); pipe_bicgstab_rr runs through its own one-launch form with replacement steps
    (krr = 5)."""
    H.lib().bicg_comm_init_single(0)
    n8 = (synth.TRANSPORT_N + 7) // 8
    A = synth.transport_like(n=n8, scale_decades=SCALE)
    coo = A.to_coo()
    ctx = H.Context(H.single_rank_blocks(A))
    fl = ctx.flags()
    assert fl["persist"] and fl["all_sell"] and fl["col16"], fl
    _spmv_check(ctx, A, coo, seed=8)
    _trajectory_check(ctx, A, coo, ("pipe_bicgstab", "bicgstab", "ca_bicgstab", "pipe_bicgstab_rr"))
    ctx.close()


@pytest.mark.parametrize("which", ["shifted_pipe_lopbicgstab", "shifted_lopbicgstab"])
def test_shifted_on_a_rank_of_8_is_one_persistent_launch(which):
    """BASELINE.json configs[4] on what ONE of 8 GPUs holds (200 264 rows, 16 shifts, seed 7): shifted_pipe_lopbicgstab and
    shifted_lopbicgstab run as persistent launches (k_shpipe_persist / k_shlop_persist: seed vectors in registers, the 15 other
    shifts' p_j / x_j streamed through while the last dot group travels, the per-shift coefficients published by the helper
    workgroup with omega) -- asserted --, the first K iterations against the oracle's trajectory, the converged solve beside the
    multi-launch form (BICG_PERSIST="shifted=0"), run-to-run bit-identical (reference src/shifted_solver.c:257-319, 794-866)."""
    import os
    H.lib().bicg_comm_init_single(0)
    n8 = (synth.TRANSPORT_N + 7) // 8
    A = synth.transport_like(n=n8, scale_decades=SCALE)
    row, col, val = A.to_coo()
    nsh, seed = 16, 7
    sigma = (np.arange(nsh) + 1.0) * 0.01 / nsh
    ones = np.ones(A.rows)
    ctx = H.Context(H.single_rank_blocks(A))
    assert ctx.flags()["persist"]
    b = ctx.spmv(ones) + sigma[seed] * ones
    orc = O.solve_shifted(A.rows, row, col, val, b, sigma, seed, tol=0.0, max_iter=K, which=which)
    got = ctx.solve_shifted(b, sigma, seed, tol=0.0, max_iter=K, check_every=K, which=which)
    assert ctx.last_shifted_persistent() and got["k"] == orc["k"] == K
    tr = ctx.trace(K)
    for key in ("alpha", "omega", "beta", "dotr"):
        np.testing.assert_allclose(tr[key], orc[key], rtol=1e-7, err_msg=f"{which} {key}")
    assert np.abs(got["x"] - orc["x"]).max() <= 1e-8 * np.abs(orc["x"]).max()
    again = ctx.solve_shifted(b, sigma, seed, tol=0.0, max_iter=K, check_every=K, which=which)
    assert np.array_equal(again["x"], got["x"]) and np.array_equal(again["r"], got["r"])
    # to a tolerance the system reaches (the oracle needs 711 iterations for 1e-10 on this matrix): every shift's true residual,
    # and the multi-launch form beside it (pipelined recurrences: the two forms round their dot sums differently and part ways
    # slowly -- same convergence, not the same bits)
    full = ctx.solve_shifted(b, sigma, seed, tol=1e-5, max_iter=600, which=which)
    assert ctx.last_shifted_persistent() and 100 < full["k"] < 600
    rel = ctx.shifted_residuals(full["x"], b, sigma)
    assert rel.max() < 1e-4, rel
    H.switches(persist_shifted=0)
    try:
        ref = H.Context(H.single_rank_blocks(A))
        multi = ref.solve_shifted(b, sigma, seed, tol=1e-5, max_iter=600, which=which)
        assert not ref.last_shifted_persistent()
        ref_rel = ref.shifted_residuals(multi["x"], b, sigma)
        ref.close()
    finally:
        H.switches(persist_shifted=None)
    # (the iterates themselves are not comparable at this tolerance: residual 1e-5 on a matrix scaled over two decades leaves
    # errors of several per cent in x -- in both forms)
    assert abs(full["k"] - multi["k"]) <= 0.1 * multi["k"], (full["k"], multi["k"])
    assert ref_rel.max() < 1e-4, ref_rel
    ctx.close()
