"""Shifted BiCGStab (reference src/shifted_solver.c, shifted_lopbicgstab and its two re-ordered
variants): oracle pinned bit-exactly to the real reference (CPU), HIP path vs oracle / golden (GPU)."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
from mpi_bicgstab_amd import synth

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "shifted_*.npz")))


def _load(path):
    g = np.load(path)
    n = int(g["n"])
    A = synth.CSR(n, n, g["ptr"], g["col"], g["val"])
    return g, A


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_bitexact_vs_reference(path):
    g, A = _load(path)
    assert bool(g["variants_bit_identical"])           # _v2 and _nooverlap == base function, bit for bit
    row, col, val = A.to_coo()
    o = O.solve_shifted(A.rows, row, col, val, g["b"], g["sigma"], int(g["seed"]))
    assert o["k"] == int(g["k"])
    assert np.array_equal(o["x"], g["x"]) and np.array_equal(o["r"], g["r"])
    assert bool(g["pipe_variants_bit_identical"])
    o = O.solve_shifted(A.rows, row, col, val, g["b"], g["sigma"], int(g["seed"]), which="shifted_pipe_lopbicgstab")
    assert o["k"] == int(g["pipe_k"])
    assert np.array_equal(o["x"], g["pipe_x"]) and np.array_equal(o["r"], g["pipe_r"])
    o = O.solve_shifted(A.rows, row, col, val, g["xi_b"], g["sigma"], 0, which="shifted_bicgstab")
    assert o["k"] == int(g["xi_k"])
    assert np.array_equal(o["x"], g["xi_x"]) and np.array_equal(o["r"], g["xi_r"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_hip_shifted_vs_golden(path):
    from mpi_bicgstab_amd import hipsolver as H
    H.lib().bicg_comm_init_single(0)
    g, A = _load(path)
    sigma, seed = g["sigma"], int(g["seed"])
    ctx = H.Context(H.single_rank_blocks(A))
    res = ctx.solve_shifted(g["b"], sigma, seed)
    assert abs(res["k"] - int(g["k"])) <= 2
    # every shifted system is solved: ||(A + sigma_j I) x_j - b|| / ||b||  (reference src/test_shifted.c:129-154)
    row, col, val = A.to_coo()
    for j in range(len(sigma)):
        resid = O.spmv(A.rows, row, col, val, res["x"][j]) + sigma[j] * res["x"][j] - g["b"]
        ref_resid = O.spmv(A.rows, row, col, val, g["x"][j]) + sigma[j] * g["x"][j] - g["b"]
        rel, rel_ref = np.linalg.norm(resid) / np.linalg.norm(g["b"]), np.linalg.norm(ref_resid) / np.linalg.norm(g["b"])
        assert rel <= max(10 * rel_ref, 1e-11), (j, rel, rel_ref)
    assert np.abs(res["x"][seed] - 1.0).max() <= 1e-9                 # manufactured seed solution
    assert np.abs(res["x"] - g["x"]).max() <= 1e-9
    # first iterations of the seed recurrence against the oracle
    o = O.solve_shifted(A.rows, row, col, val, g["b"], sigma, seed)
    tr = ctx.trace(res["k"])
    h = min(6, res["k"], o["k"])
    for key in ("alpha", "omega", "beta", "dotr"):
        np.testing.assert_allclose(tr[key][:h], o[key][:h], rtol=1e-8, err_msg=key)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["shifted_pipe_lopbicgstab", "shifted_bicgstab"])
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_hip_shifted_variants_vs_golden(path, which):
    from mpi_bicgstab_amd import hipsolver as H
    H.lib().bicg_comm_init_single(0)
    g, A = _load(path)
    sigma = g["sigma"]
    pre = "pipe_" if which == "shifted_pipe_lopbicgstab" else "xi_"
    seed = int(g["seed"]) if which == "shifted_pipe_lopbicgstab" else 0
    b = g["b"] if which == "shifted_pipe_lopbicgstab" else g["xi_b"]
    ctx = H.Context(H.single_rank_blocks(A))
    res = ctx.solve_shifted(b, sigma, seed, which=which)
    assert abs(res["k"] - int(g[pre + "k"])) <= 2
    assert np.abs(res["x"] - g[pre + "x"]).max() <= 1e-8 * max(1.0, np.abs(g[pre + "x"]).max())
    row, col, val = A.to_coo()
    shift0 = 0.0 if which == "shifted_bicgstab" else None
    for j in range(len(sigma)):
        sg = sigma[j] if not (which == "shifted_bicgstab" and j == 0) else 0.0     # seed system of shifted_bicgstab is A
        resid = O.spmv(A.rows, row, col, val, res["x"][j]) + sg * res["x"][j] - b
        ref_resid = O.spmv(A.rows, row, col, val, g[pre + "x"][j]) + sg * g[pre + "x"][j] - b
        rel, rel_ref = np.linalg.norm(resid) / np.linalg.norm(b), np.linalg.norm(ref_resid) / np.linalg.norm(b)
        assert rel <= max(10 * rel_ref, 1e-10), (j, rel, rel_ref)
    o = O.solve_shifted(A.rows, row, col, val, b, sigma, seed, which=which)
    tr = ctx.trace(res["k"])
    h = min(6, res["k"], o["k"])
    for key in ("alpha", "omega", "beta", "dotr"):
        np.testing.assert_allclose(tr[key][:h], o[key][:h], rtol=1e-8, err_msg=key)
    ctx.close()


@pytest.mark.gpu
def test_hip_shifted_single_shift_equals_plain():
    """one shift, seed 0, sigma = 0: the shifted recurrence reduces to plain BiCGStab on A"""
    from mpi_bicgstab_amd import hipsolver as H
    H.lib().bicg_comm_init_single(0)
    A = synth.stencil7(9)
    ctx = H.Context(H.single_rank_blocks(A))
    b = ctx.spmv(np.ones(A.rows))
    sh = ctx.solve_shifted(b, np.array([0.0]), 0, tol=1e-12)
    pl = ctx.solve("bicgstab", b, tol=1e-12)
    assert abs(sh["k"] - pl["k"]) <= 1
    assert np.abs(sh["x"][0] - 1.0).max() <= 1e-9
    ctx.close()


@pytest.mark.gpu
def test_spmm_columns_are_the_spmv_of_each_vector():
    """bicg_spmm / the SpMM inside bicg_shifted_residuals (A read once for 16 shifts; the reference's verification loop
    src/test_shifted.c:129-154 does one SpMV per shift): every column bit-identical to the single-vector SpMV with the
    shift added the reference's way, residual norms equal to the per-shift loop's."""
    from mpi_bicgstab_amd import hipsolver as H
    H.lib().bicg_comm_init_single(0)
    A = synth.from_offsets(30011, (0, 1, -1, 37, -37, 2999, -2999), diag_base=9.0, seed=5)
    ctx = H.Context(H.single_rank_blocks(A))
    H.switches(spmm_window=0)                 # the row-major kernel (round 2) beside the windowed one (round 4)
    try:
        rowmajor = H.Context(H.single_rank_blocks(A))
        Xr = np.random.default_rng(3).standard_normal((16, A.rows))
        Yr, _ = rowmajor.spmm(Xr, 0.01 * (np.arange(16) + 1.0))
        assert not rowmajor.last_spmm_windowed()
        rowmajor.close()
    finally:
        H.switches(spmm_window=None)
    H.switches(spmm_window=1)                 # the windowed kernel (round 4: x staged through registers, one group per workgroup)
    try:
        windowed = H.Context(H.single_rank_blocks(A))
        Yw1, _ = windowed.spmm(Xr, 0.01 * (np.arange(16) + 1.0))
        assert windowed.last_spmm_kind() == "windowed" and np.array_equal(Yw1, Yr)
        windowed.close()
    finally:
        H.switches(spmm_window=None)
    Yw, _ = ctx.spmm(Xr, 0.01 * (np.arange(16) + 1.0))
    # three clusters of offsets: the pipelined kernel (round 6, csrc/bicg_spmm.hip: windows copied by the DMA path a step ahead)
    assert ctx.last_spmm_kind() == "pipelined" and np.array_equal(Yw, Yr)
    rng = np.random.default_rng(8)
    for nvec in (1, 5, 16, 21):                       # less than, exactly and more than one pass of 16
        X = rng.standard_normal((nvec, A.rows))
        sigma = 0.01 * (np.arange(nvec) + 1.0)
        Y, ms = ctx.spmm(X, sigma)
        for j in range(nvec):
            yj = ctx.spmv(X[j])
            yj = yj + sigma[j] * X[j]                  # my_daxpy(sigma_j, x_j, y_j): one rounding per element, like the kernel
            assert np.array_equal(Y[j], yj), (nvec, j)
        Y0, _ = ctx.spmm(X)                            # no shift: plain A X
        assert np.array_equal(Y0[0], ctx.spmv(X[0]))
        b = rng.standard_normal(A.rows)
        r1 = ctx.shifted_residuals(X, b, sigma)
        H.switches(spmm=0)
        try:
            r2 = ctx.shifted_residuals(X, b, sigma)    # one SpMV + one fused norm kernel per shift
        finally:
            H.switches(spmm=None)
        np.testing.assert_allclose(r1, r2, rtol=1e-13)
        want = [np.linalg.norm(b - (ctx.spmv(X[j]) + sigma[j] * X[j])) / np.linalg.norm(b) for j in range(nvec)]
        np.testing.assert_allclose(r1, want, rtol=1e-12)
    ctx.close()


@pytest.mark.gpu
def test_hip_shifted_dropin_symbols():
    """The reference call surface of src/shifted_solver.h:16-21 itself -- the six functions on host CSR blocks and host
    vectors, x_loc_set laid out [shift][row] -- not only the handle API the other tests use: every symbol returns the
    iteration count and the bits of the corresponding bicg_solve_shifted call (same kernels underneath), the re-ordered
    spellings of one algorithm agree with each other like the reference's do (fixtures: variants_bit_identical), and the
    solution is the reference's to 1e-9."""
    import ctypes as C
    from mpi_bicgstab_amd import hipsolver as H
    H.lib().bicg_comm_init_single(0)
    g, A = _load(GOLDEN[0])          # 3001 rows, 16 shifts, seed 7
    sigma = np.ascontiguousarray(g["sigma"], dtype=np.float64)
    seed = int(g["seed"])
    ctx = H.Context(H.single_rank_blocks(A))
    want = {"lop": ctx.solve_shifted(g["b"], sigma, seed, which="shifted_lopbicgstab"),
            "pipe": ctx.solve_shifted(g["b"], sigma, seed, which="shifted_pipe_lopbicgstab"),
            "xi": ctx.solve_shifted(g["xi_b"], sigma, 0, which="shifted_bicgstab")}
    ctx.close()
    blocks = H.single_rank_blocks(A)
    dp = C.POINTER(C.c_double)
    calls = [("shifted_lopbicgstab", "lop", "b", True), ("shifted_lopbicgstab_v2", "lop", "b", True),
             ("shifted_lopbicgstab_nooverlap", "lop", "b", True), ("shifted_pipe_lopbicgstab", "pipe", "b", True),
             ("shifted_pipe_lopbicgstab_nooverlap", "pipe", "b", True), ("shifted_bicgstab", "xi", "xi_b", False)]
    os.environ["BICG_QUIET"] = "1"
    try:
        for name, key, rhs, has_seed in calls:
            fn = getattr(H.lib(), name)
            fn.restype = C.c_int
            x = np.zeros(len(sigma) * A.rows)
            r = np.array(g[rhs], dtype=np.float64)
            args = [C.byref(blocks.diag), C.byref(blocks.offd), C.byref(blocks.info), x.ctypes.data_as(dp), r.ctypes.data_as(dp),
                    sigma.ctypes.data_as(dp), C.c_int(len(sigma))]
            if has_seed:
                args.append(C.c_int(seed))
            k = fn(*args)
            w = want[key]
            assert k == w["k"], (name, k, w["k"])
            assert np.array_equal(x.reshape(len(sigma), A.rows), w["x"]) and np.array_equal(r, w["r"]), name
    finally:
        os.environ.pop("BICG_QUIET", None)
        H.lib().bicg_dropin_release()
    pre = {"lop": "", "pipe": "pipe_", "xi": "xi_"}
    for key, w in want.items():
        ref = g[pre[key] + "x"]
        assert abs(w["k"] - int(g[pre[key] + "k"])) <= 2, key
        assert np.abs(w["x"] - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), key
