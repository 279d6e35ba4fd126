"""Seed-switching shifted solvers (reference src/shifted_switching_solver.c):
shifted_lopbicg (per-shift stop flags), shifted_lopbicg_switching and its _noovlp twin.
CPU: the oracle restatement is pinned bit-exactly to the real reference through the committed
fixtures (tests/golden/switching_*.npz, made by make_golden_switching.py from
oracle/_ref/libref_switching.so). GPU: the HIP path against oracle / fixtures."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "switching_*.npz")))


def _coo(g):
    n = int(g["n"])
    row = np.repeat(np.arange(n, dtype=np.uint32), np.diff(g["ptr"].astype(np.int64)))
    return n, row, g["col"], g["val"]


def test_fixtures_present():
    assert len(GOLDEN) >= 4
    assert all(bool(np.load(p)["noovlp_bit_identical"]) for p in GOLDEN)     # _noovlp is an arithmetic twin
    assert sum(int(np.load(p)["sw_k"]) - 1 != int(np.load(p)["flag_k"]) for p in GOLDEN) >= 2   # switches change the course


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_bitexact_vs_reference(path):
    g = np.load(path)
    n, row, col, val = _coo(g)
    flag = O.solve_switching(n, row, col, val, g["b"], g["sigma"], int(g["seed"]), which="shifted_lopbicg")
    assert flag["k"] == int(g["flag_k"])
    assert np.array_equal(flag["x"], g["flag_x"]) and np.array_equal(flag["r"], g["flag_r"])
    assert flag["stop"].all()
    sw = O.solve_switching(n, row, col, val, g["b"], g["sigma"], int(g["seed"]), which="shifted_lopbicg_switching")
    assert sw["k"] == int(g["sw_k"])
    assert np.array_equal(sw["x"], g["sw_x"]) and np.array_equal(sw["r"], g["sw_r"])
    assert sw["stop"].all()


@pytest.mark.parametrize("P", [2, 3])
def test_oracle_switching_virtual_ranks(P):
    """the distributed restatement (P virtual ranks) follows the same course as P = 1: same number of
    switches and final seed, iteration counts within 1, solutions to 1e-9"""
    g = np.load([p for p in GOLDEN if "lin8" in p][0])
    n, row, col, val = _coo(g)
    one = O.solve_switching(n, row, col, val, g["b"], g["sigma"], int(g["seed"]))
    many = O.solve_switching(n, row, col, val, g["b"], g["sigma"], int(g["seed"]), nranks=P)
    assert one["switches"] == many["switches"] >= 1 and one["final_seed"] == many["final_seed"]
    assert abs(one["k"] - many["k"]) <= 1
    assert np.abs(one["x"] - many["x"]).max() <= 1e-9 * np.abs(one["x"]).max()


# ------------------------------------------------------------------------------------------ GPU
def _ctx(g):
    from mpi_bicgstab_amd import hipsolver as H, synth
    H.lib().bicg_comm_init_single(0)
    A = synth.CSR(int(g["n"]), int(g["n"]), g["ptr"].astype(np.uint32), g["col"].astype(np.uint32), g["val"].astype(np.float64))
    return H.Context(H.single_rank_blocks(A)), A


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["shifted_lopbicg", "shifted_lopbicg_switching"])
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_hip_switching_vs_reference_fixture(path, which):
    """HIP path against the REAL reference's outputs. SpMV rows, vector updates and all per-shift
    scalar recurrences use the reference's operation order; only the dot products associate
    differently, so the course (iteration count, stop flags, the seed switch) is the same and the
    solutions agree to the solver tolerance."""
    g = np.load(path)
    ctx, A = _ctx(g)
    key = "flag" if which == "shifted_lopbicg" else "sw"
    sigma, seed = g["sigma"], int(g["seed"])
    got = ctx.solve_shifted(g["b"], sigma, seed, which=which, tol=1e-12, check_every=7)
    n, row, col, val = _coo(g)
    orc = O.solve_switching(n, row, col, val, g["b"], sigma, seed, which=which)
    # iteration count: inside the reference's OWN spread over rank counts (a 90-iteration BiCGStab
    # course is sensitive to the association of the dot sums: the oracle gives 85..96 on the
    # transport fixture for P = 1..8), +-2
    ks = [int(g[key + "_k"])] + [O.solve_switching(n, row, col, val, g["b"], sigma, seed, which=which, nranks=P)["k"] for P in (2, 3, 4, 8)]
    assert min(ks) - 2 <= got["k"] <= max(ks) + 2, (got["k"], ks)
    assert got["switches"] == orc["switches"]
    assert got["iterations"] == got["k"] - (1 if which == "shifted_lopbicg_switching" else 0)
    ref_x = g[key + "_x"]
    scale = np.abs(ref_x).max()
    assert np.abs(got["x"] - ref_x).max() <= 2e-9 * scale
    # every system is solved: || (A + sigma_j I) x_j - b || <= 1e-10 ||b||   (check of src/test_shifted.c:129-154)
    rel = ctx.shifted_residuals(got["x"], g["b"], sigma)        # computed on the device
    assert rel.max() <= 1e-10, rel
    j = len(sigma) // 2                                          # ... and cross-checked on the host for one shift
    res = ctx.spmv(got["x"][j]) + sigma[j] * got["x"][j] - g["b"]
    assert abs(np.linalg.norm(res) / np.linalg.norm(g["b"]) - rel[j]) <= 1e-13
    # the seed system's recurrence scalars follow the oracle's for the first iterations
    tr = ctx.trace(got["iterations"])
    h = min(6, got["iterations"], len(orc["dotr"]))
    np.testing.assert_allclose(tr["dotr"][:h], orc["dotr"][:h], rtol=1e-8)
    np.testing.assert_allclose(tr["alpha"][:h], orc["alpha"][:h], rtol=1e-8)
    ctx.close()


@pytest.mark.gpu
def test_hip_switching_host_check_interval_does_not_matter():
    """the device pauses itself at a seed switch: iterating in chunks of 1 or 50 gives the same bits"""
    g = np.load([p for p in GOLDEN if "transport" in p][0])
    ctx, A = _ctx(g)
    a = ctx.solve_shifted(g["b"], g["sigma"], int(g["seed"]), which="shifted_lopbicg_switching", tol=1e-12, check_every=1)
    b = ctx.solve_shifted(g["b"], g["sigma"], int(g["seed"]), which="shifted_lopbicg_switching", tol=1e-12, check_every=50)
    assert a["k"] == b["k"] and a["switches"] == b["switches"] >= 1
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["r"], b["r"])
    ctx.close()


@pytest.mark.gpu
def test_hip_switching_dropin_symbols():
    """the reference call surface of src/shifted_switching_solver.h:10-12 on host arrays"""
    import ctypes as C
    from mpi_bicgstab_amd import hipsolver as H
    g = np.load([p for p in GOLDEN if "lin8" in p][0])
    ctx, A = _ctx(g)
    ctx.close()
    blocks = H.single_rank_blocks(A)
    sigma = np.ascontiguousarray(g["sigma"], dtype=np.float64)
    outs = {}
    for name in ("shifted_lopbicg", "shifted_lopbicg_switching", "shifted_lopbicg_switching_noovlp"):
        fn = getattr(H.lib(), name)
        fn.restype = C.c_int
        x = np.zeros(len(sigma) * A.rows)
        r = np.array(g["b"], dtype=np.float64)
        dp = C.POINTER(C.c_double)
        os.environ["BICG_QUIET"] = "1"
        try:
            k = fn(C.byref(blocks.diag), C.byref(blocks.offd), C.byref(blocks.info), x.ctypes.data_as(dp), r.ctypes.data_as(dp),
                   sigma.ctypes.data_as(dp), C.c_int(len(sigma)), C.c_int(int(g["seed"])))
        finally:
            os.environ.pop("BICG_QUIET", None)
        outs[name] = (k, x)
    assert abs(outs["shifted_lopbicg"][0] - int(g["flag_k"])) <= 2
    assert abs(outs["shifted_lopbicg_switching"][0] - int(g["sw_k"])) <= 2
    assert outs["shifted_lopbicg_switching"][0] == outs["shifted_lopbicg_switching_noovlp"][0]
    assert np.array_equal(outs["shifted_lopbicg_switching"][1], outs["shifted_lopbicg_switching_noovlp"][1])
    ref = g["sw_x"].reshape(-1)
    assert np.abs(outs["shifted_lopbicg_switching"][1] - ref).max() <= 2e-9 * np.abs(ref).max()


# ------------------------------------------------------------------------------------------
# P > 1: the REAL reference under mpiexec (tests/golden/ranks_*.npz, make_golden_shifted_ranks.py)
RANKS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ranks_*.npz")))


@pytest.mark.parametrize("P", [2, 4])
@pytest.mark.parametrize("path", RANKS, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_shifted_family_bitexact_at_P_ranks(path, P):
    """Every shifted solver of the oracle, run over P virtual ranks, reproduces the real reference at P MPI
    ranks bit for bit: shifted_lopbicgstab, shifted_pipe_lopbicgstab, shifted_bicgstab
    (src/shifted_solver.c) and shifted_lopbicg, shifted_lopbicg_switching (src/shifted_switching_solver.c)."""
    g = np.load(path)
    n, row, col, val = _coo(g)
    sigma, seed = g["sigma"], int(g["seed"])
    assert len(RANKS) >= 2
    for fn in ("shifted_lopbicgstab", "shifted_pipe_lopbicgstab", "shifted_bicgstab"):
        b = g[f"{fn}_P{P}_b"]
        got = O.solve_shifted(n, row, col, val, b, sigma, 0 if fn == "shifted_bicgstab" else seed, nranks=P, which=fn)
        assert got["k"] == int(g[f"{fn}_P{P}_k"]), fn
        # equal_nan: on the stencil case the reference's pipelined variant breaks down into NaNs at P = 2
        # (102 iterations) -- and so does the restatement, at the same iteration
        assert np.array_equal(got["x"], g[f"{fn}_P{P}_x"], equal_nan=True), fn
        assert np.array_equal(got["r"], g[f"{fn}_P{P}_r"], equal_nan=True), fn
    for fn in ("shifted_lopbicg", "shifted_lopbicg_switching"):
        b = g[f"{fn}_P{P}_b"]
        got = O.solve_switching(n, row, col, val, b, sigma, seed, nranks=P, which=fn)
        assert got["k"] == int(g[f"{fn}_P{P}_k"]), fn
        assert np.array_equal(got["x"], g[f"{fn}_P{P}_x"]) and np.array_equal(got["r"], g[f"{fn}_P{P}_r"]), fn
