"""`-m gpu`: the unstructured FEM matrix (mpi_bicgstab_amd.mesh, 1 601 613 rows, 26.0 M non-zeros -- the stand-in for
Transport.mtx, reference README.md:32-42) in its three numberings on one rank: which product kernel each gets (asserted), y = A x
bit for bit against the oracle (reference src/matrix.c:498-516), the first 8 iterations of the four solvers against the oracle's
alpha / omega / beta / (r,r) (src/solver.c:35-576).

Horizon and tolerance: on this matrix (slivers, two decades of scaling) the REFERENCE's own scalars move by a factor of ten per
iteration when only the association of its dot sums changes -- P = 1 against P = 2 ranks of the oracle: alpha differs by 7e-14 at
iteration 1, 2e-9 at iteration 8, 4.5e-4 at iteration 12 (beta 2.5e-3). Eight iterations at rtol 1e-6 is what a trajectory
comparison can mean here; the products themselves are compared bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
from mpi_bicgstab_amd import hipsolver as H
from mpi_bicgstab_amd import mesh

pytestmark = pytest.mark.gpu

K_FIX = 8
# the kernel a numbering must get (hipsolver.product_kernels): generator order -- a few long runs of columns per 256-row group -- the
# three-trip product with the x window in LDS staged run by run; reverse Cuthill-McKee (59-170 short runs, <= 1 700 distinct columns
# per group) the same product with a LIST-driven window (every distinct column once, 16 bits each); the random permutation (4 500
# distinct columns per group: no window) the three-trip product that gathers x through the caches by 32-bit columns
EXPECT = {"generator": ("jagw", dict(window=True, col16=True)), "rcm": ("jagw_list", dict(window=True, col16=True)),
          "random": ("jagd", dict(window=False, col16=False))}


@pytest.fixture(scope="module", params=["generator", "rcm", "random"])
def case(request, tmp_path_factory):
    H.lib().bicg_comm_init_single(0)
    cache = tmp_path_factory.getbasetemp() / "mesh_cache"
    cache.mkdir(exist_ok=True)
    A = mesh.fem_unstructured(117, request.param, scale_decades=2.0, cache_dir=str(cache))
    ctx = H.Context(H.single_rank_blocks(A))
    yield request.param, A, A.to_coo(), ctx
    ctx.close()


def test_kernel_and_plan(case):
    kind, A, _, ctx = case
    assert A.rows == 117 ** 3 and 25_500_000 < A.nnz < 26_500_000
    fl, info = ctx.flags(), ctx.plan_info()
    kernel, want = EXPECT[kind]
    assert fl["jagged"] and fl["all_sell"] and info["sell_rows"] == A.rows and info["sell_padding"] == 0, (fl, info)
    for k, v in want.items():
        assert fl[k] == v, (kind, k, fl)
    H.product_kernels()
    ctx.spmv_bench(3)
    assert H.product_kernels() == [kernel], kind
    # the stored layout: 8-byte values + 2-byte offsets or slots (4-byte columns for the random numbering), no padding
    per_nnz = ctx.spmv_matrix_bytes() / A.nnz
    assert per_nnz < (12.5 if kind == "random" else 11.4 if kind == "rcm" else 10.6), per_nnz     # (rcm: + 0.7 B per non-zero of window lists)
    if kind == "rcm":       # ... and without the list: the gathers through the caches (what ranks of a multi-rank run keep)
        H.switches(window_list=0)
        try:
            plain = H.Context(H.single_rank_blocks(A))
            assert not plain.flags()["window"] and plain.flags()["col16"]
            H.product_kernels()
            plain.spmv_bench(3)
            assert H.product_kernels() == ["jagd"]
            x = np.cos(np.arange(A.rows))
            assert np.array_equal(plain.spmv(x), ctx.spmv(x))          # the two products agree bit for bit
            plain.close()
        finally:
            H.switches(window_list=None)


def test_spmv_bitexact_and_linear(case):
    kind, A, (row, col, val), ctx = case
    rng = np.random.default_rng(5)
    x, z = rng.standard_normal(A.rows), rng.standard_normal(A.rows)
    y = ctx.spmv(x)
    assert np.array_equal(y, O.spmv(A.rows, row, col, val, x)), kind          # 1.6 M ragged rows, bit for bit
    lhs = ctx.spmv(2.5 * x - 0.75 * z)
    rhs = 2.5 * y - 0.75 * ctx.spmv(z)
    scale = np.abs(A.val).max() * 30 * (np.abs(x).max() + np.abs(z).max())
    assert np.abs(lhs - rhs).max() <= 1e-13 * scale


def test_first_iterations_against_the_oracle(case):
    kind, A, (row, col, val), ctx = case
    b = ctx.spmv(np.ones(A.rows))
    assert np.array_equal(b, O.spmv(A.rows, row, col, val, np.ones(A.rows)))
    for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"):
        H.product_kernels()
        got = ctx.solve(method, b, tol=0.0, max_iter=K_FIX, krr=5, nrr=1, check_every=K_FIX)
        orc = O.solve(method, A.rows, row, col, val, b, tol=0.0, max_iter=K_FIX, krr=5, nrr=1)
        assert got["k"] == orc["k"] == K_FIX
        assert EXPECT[kind][0] in H.product_kernels(), (kind, method)
        tr = ctx.trace(K_FIX)
        for key in ("alpha", "omega", "beta", "dotr"):
            np.testing.assert_allclose(tr[key], orc[key], rtol=1e-6, err_msg=f"{kind} {method} {key}")
        assert np.abs(got["x"] - orc["x"]).max() <= 1e-6 * np.abs(orc["x"]).max(), (kind, method)


def test_shifted_solve_on_the_list_driven_window(case):
    """the shifted products (A + sigma_seed I) x of src/shifted_solver.c:259-260 through the same kernels: 4 shifts, 6 iterations of
    shifted_lopbicgstab against the oracle's seed scalars and two of its solutions (RCM numbering: the list-driven window)"""
    kind, A, (row, col, val), ctx = case
    if kind != "rcm":
        pytest.skip("one numbering is enough: the shift rides in the product's epilogue whatever the layout")
    sigma, seed = np.array([0.01, 0.02, 0.03, 0.04]), 1
    b = ctx.spmv(np.ones(A.rows)) + sigma[seed] * np.ones(A.rows)
    H.product_kernels()
    got = ctx.solve_shifted(b, sigma, seed, tol=0.0, max_iter=6, check_every=6, which="shifted_lopbicgstab")
    assert "jagw_list" in H.product_kernels()
    orc = O.solve_shifted(A.rows, row, col, val, b, sigma, seed, tol=0.0, max_iter=6, which="shifted_lopbicgstab")
    assert got["k"] == orc["k"] == 6
    tr = ctx.trace(6)
    for key in ("alpha", "omega", "beta", "dotr"):
        np.testing.assert_allclose(tr[key], orc[key], rtol=1e-7, err_msg=key)
    for j in (0, 3):
        assert np.abs(got["x"][j] - orc["x"][j]).max() <= 1e-8 * np.abs(orc["x"][j]).max(), j


def test_spmm_on_the_unstructured_matrix(case):
    """Y_j = (A + sigma_j I) X_j for 16 vectors with the matrix read once (the reference's verification loop, src/test_shifted.c:
    129-154) on layouts with an x window: generator order (runs) and RCM (round 6: the window read from the group's column list --
    before, this layout had no SpMM and took one product per shift), both through the pipeline for ragged rows (k_spmm_jpipe,
    csrc/bicg_spmm_jag.hip) and, with BICG_PLAN=spmm-window=1, through k_spmm_win: the same bits. Every column bit for bit the single
    product + shift, 16 vectors and 5 (an odd number: the last step holds one); the residual norms of bicg_shifted_residuals against
    numpy on those columns. The random permutation (no window) keeps the row-major kernel: same bits."""
    kind, A, (row, col, val), ctx = case
    rng = np.random.default_rng(16)
    X = rng.standard_normal((16, A.rows))
    sg = (np.arange(16) + 1.0) * 0.01 / 16
    assert ctx.flags()["spmm"], ctx.flags()
    Y, ms = ctx.spmm(X, sg)
    assert ctx.last_spmm_kind() == ("rowmajor" if kind == "random" else "pipelined"), (kind, ctx.last_spmm_kind())
    for j in (0, 7, 15):
        assert np.array_equal(Y[j], ctx.spmv(X[j]) + sg[j] * X[j]), (kind, j)
    Y5, _ = ctx.spmm(X[:5], sg[:5])
    assert np.array_equal(Y5, Y[:5]), kind
    Y0, _ = ctx.spmm(X[:3])                                    # no shifts: plain A X
    assert np.array_equal(Y0[2], ctx.spmv(X[2])), kind
    if kind != "random":
        H.switches(spmm_window=1)                              # (read when a context's SpMM buffers are set up: a second context)
        try:
            other = H.Context(H.single_rank_blocks(A))
            Yw, _ = other.spmm(X, sg)
            assert other.last_spmm_kind() == "windowed" and np.array_equal(Yw, Y), kind
            other.close()
        finally:
            H.switches(spmm_window=None)
    b = rng.standard_normal(A.rows)
    got = np.asarray(ctx.shifted_residuals(X, b, sg))
    want = np.array([np.linalg.norm(b - Y[j]) / np.linalg.norm(b) for j in range(16)])
    np.testing.assert_allclose(got, want, rtol=1e-12)
