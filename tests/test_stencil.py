"""The plane-marching product (csrc/bicg_stencil.hip, `-m gpu`): blocks in whose lists the plan finds the 7-point stencil of a
grid (BASELINE.json configs[3]) are multiplied by wavefronts that march through the planes of their own grid lines. Same sums in
the same order as mult() (reference src/matrix.c:506-515): every product here is compared bit for bit with the CPU oracle and
with the slice-by-slice product (BICG_PLAN="stencil=0"); the dot sums are associated differently (another tiling), so the solvers'
scalars agree to rounding, not in bits. CA-BiCGStab's q / y phase (reference src/solver.c:225-232) in the epilogue of z = A s:
same expressions, checked against the unfused iteration."""
import numpy as np
import pytest

import oracle_lib as O
from mpi_bicgstab_amd import hipsolver as H
from mpi_bicgstab_amd import synth

pytestmark = pytest.mark.gpu

W = (6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.3)          # every position its own weight: a swapped neighbour shows
W2 = (7.25, -0.7, -1.4, -0.6, -1.5, -0.9, -1.1)

KNOBS = ("stencil", "lines", "planes", "ca_fuse", "wide")          # tokens of BICG_PLAN (csrc/bicg_knobs.h)


def _ctx(monkeypatch, A, **env):
    H.switches(**{k: env.get(k) for k in KNOBS})
    return H.Context(H.single_rank_blocks(A))


@pytest.mark.parametrize("shape,kw", [
    ((64, 8, 5), {}),                                        # one x segment, an odd number of planes
    ((128, 12, 7), {}),                                      # two x segments, 12 lines: three wavefronts of a workgroup have work
    ((192, 4, 9), {"upper_weights": W2}),                    # two value lists for the interior's distances
    ((64, 6, 6), {"wrap_y": True}),                          # a plane's first / last line keeps its -sy / +sy entry
    ((256, 16, 16), {"upper_weights": W2, "wrap_y": True}),
    ((256, 8, 6), {"wrap_y": True}),                         # one wavefront line of four segments, both face segments in it
    ((512, 8, 5), {}),                                       # BASELINE.json configs[3]'s line length
    ((384, 4, 4), {}),                                       # six segments: pairs, not fours
])
def test_plane_marching_product_bit_for_bit(monkeypatch, shape, kw):
    H.lib().bicg_comm_init_single(0)
    nx, ny, nz = shape
    A = synth.grid7(nx, ny, nz, W, **kw)
    row, col, val = A.to_coo()
    x = np.random.default_rng(nx + ny + nz).standard_normal(A.rows)
    want = O.spmv(A.rows, row, col, val, x)
    tiles = [{}, {"lines": 2, "planes": 3}, {"lines": 2, "planes": 64},
             {"lines": 4, "planes": 1}, {"lines": 4, "planes": 5}]
    # the wide form (2 / 4 rows per lane): taken when the lines are whole groups of 2 / 4 x segments and the block has one value
    # per distance -- two value lists (upper_weights) keep one row per lane whatever is asked for
    tiles += [dict(t, wide=w) for w in (2, 4) for t in ({"lines": 2, "planes": 3}, {"lines": 4, "planes": 5}, {"lines": 2, "planes": 64})]
    for env in tiles:
        if env.get("lines") == 4 and ny % 4:
            continue
        ctx = _ctx(monkeypatch, A, **env)
        info = ctx.stencil_info()
        assert info["on"] == 1 and info["sy"] == nx and info["ny"] == ny and info["nz"] == nz, (env, info)
        if "lines" in env:
            assert info["lines"] == env["lines"] and info["planes"] == env["planes"]
        wide = env.get("wide", 1)
        assert info["rows_per_lane"] == (wide if (nx // 64) % wide == 0 and "upper_weights" not in kw else 1), (env, info)
        assert np.array_equal(ctx.spmv(x), want), env
        assert np.array_equal(ctx.spmv(x), want), env             # the reversed direction of the second product
        ctx.close()
    ctx = _ctx(monkeypatch, A, stencil=0)
    assert ctx.stencil_info()["on"] == 0
    assert np.array_equal(ctx.spmv(x), want)
    ctx.close()


@pytest.mark.parametrize("shape", [(96, 8, 8), (64, 7, 5)])
def test_grids_the_product_does_not_take(monkeypatch, shape):
    """lines that are not a multiple of 64 rows, an odd number of lines per plane: the slice-by-slice product"""
    H.lib().bicg_comm_init_single(0)
    A = synth.grid7(*shape, W)
    row, col, val = A.to_coo()
    x = np.random.default_rng(5).standard_normal(A.rows)
    ctx = _ctx(monkeypatch, A)
    assert ctx.stencil_info()["on"] == 0
    assert np.array_equal(ctx.spmv(x), O.spmv(A.rows, row, col, val, x))
    ctx.close()


def _traces(ctx, b, methods=("bicgstab", "ca_bicgstab", "pipe_bicgstab"), k=12):
    out = []
    for method in methods:
        res = ctx.solve(method, b, tol=0.0, max_iter=k, check_every=k)
        tr = ctx.trace(k)
        out.append((np.concatenate([tr[key] for key in ("alpha", "omega", "beta", "dotr")]), res["x"].copy()))
    return out


def test_solvers_on_the_plane_marching_product(monkeypatch):
    """12 iterations of the three solvers: the scalars and x of the plane-marching product (its own tiling of the dot sums) against
    the slice-by-slice product's to rounding -- 1e-8 relative after 12 iterations of a well-conditioned system --, CA-BiCGStab
    with q / y in the product's epilogue against the five-kernel iteration, and against the oracle's trace."""
    H.lib().bicg_comm_init_single(0)
    monkeypatch.setenv("BICG_PERSIST", "0")
    A = synth.grid7(128, 16, 12, W)
    row, col, val = A.to_coo()
    b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
    runs = {}
    for name, env in (("slices", {"stencil": 0}), ("planes", {}), ("planes_unfused", {"ca_fuse": 0}),
                      ("planes_2x3", {"lines": 2, "planes": 3}), ("planes_wide", {"wide": 2}),
                      ("planes_wide_unfused", {"wide": 2, "ca_fuse": 0, "lines": 2, "planes": 5})):
        ctx = _ctx(monkeypatch, A, **env)
        assert ctx.stencil_info()["on"] == (0 if name == "slices" else 1)
        assert ctx.stencil_info()["rows_per_lane"] == (0 if name == "slices" else 2 if "wide" in name else 1)
        runs[name] = _traces(ctx, b)
        ctx.close()
    for name in ("planes", "planes_unfused", "planes_2x3", "planes_wide", "planes_wide_unfused"):
        for (t, x), (t0, x0) in zip(runs[name], runs["slices"]):
            np.testing.assert_allclose(t, t0, rtol=1e-8, atol=0.0)
            np.testing.assert_allclose(x, x0, rtol=1e-8, atol=1e-12)
    for i, method in enumerate(("bicgstab", "ca_bicgstab", "pipe_bicgstab")):
        orc = O.solve(method, A.rows, row, col, val, b, tol=0.0, max_iter=12)
        want = np.concatenate([orc[key][:12] for key in ("alpha", "omega", "beta", "dotr")])
        np.testing.assert_allclose(runs["planes"][i][0], want, rtol=1e-8, atol=0.0)      # KAT-2 of SURVEY.md section 8c


def test_ca_epilogue_leaves_the_vectors_of_the_unfused_iteration(monkeypatch):
    """one CA-BiCGStab iteration, then the true residual: x and the recursive r of the fused iteration satisfy r = b - A x as
    well as the unfused iteration's do, and both runs end on the same x to rounding (reference src/solver.c:217-251)"""
    H.lib().bicg_comm_init_single(0)
    monkeypatch.setenv("BICG_PERSIST", "0")
    A = synth.grid7(64, 32, 20, W, upper_weights=W2)
    row, col, val = A.to_coo()
    b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
    res = {}
    for fuse in (1, 0):
        ctx = _ctx(monkeypatch, A, ca_fuse=fuse)
        assert ctx.stencil_info()["on"] == 1
        out = ctx.solve("ca_bicgstab", b, tol=0.0, max_iter=5, check_every=5)
        res[fuse] = (out["x"].copy(), out["r"].copy())
        ctx.close()
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-11, atol=1e-13)
    nb = np.linalg.norm(b)
    for fuse in (1, 0):
        true_r = b - O.spmv(A.rows, row, col, val, res[fuse][0])
        assert np.linalg.norm(true_r - res[fuse][1]) <= 1e-12 * nb
