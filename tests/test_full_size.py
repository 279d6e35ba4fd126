"""BASELINE.json's full size on the GPU (`-m gpu`): the Transport-shaped matrix (1 602 111 rows,
23 921 209 non-zeros) -- SpMV bit-exact against the oracle, size-independent properties (linearity,
true vs recursive residual), and the first iterations of every solver against the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from mpi_bicgstab_amd import hipsolver as H
from mpi_bicgstab_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    H.lib().bicg_comm_init_single(0)
    A = synth.transport_like(scale_decades=2.0)
    row, col, val = A.to_coo()
    ctx = H.Context(H.single_rank_blocks(A))
    yield A, (row, col, val), ctx
    ctx.close()


def test_plan_is_sliced_ell(big):
    A, _, ctx = big
    info = ctx.plan_info()
    assert info["sell_rows"] == A.rows                   # banded: every group on the sliced-ELL path
    assert info["sell_padding"] <= 0.01 * A.nnz
    # device footprint: sliced-ELL values + 16-bit columns + row pointers only (the CSR copy and the 32-bit
    # sliced-ELL columns no kernel reads are not uploaded): below the 12 B per non-zero of the reference's CSR
    assert ctx.device_matrix_bytes() <= 1.3 * (12 * A.nnz + 4 * (A.rows + 1))
    assert ctx.device_matrix_bytes() <= 11.5 * A.nnz


def test_spmv_bitexact_and_linear(big):
    A, (row, col, val), ctx = big
    rng = np.random.default_rng(0)
    x, z = rng.standard_normal(A.rows), rng.standard_normal(A.rows)
    y = ctx.spmv(x)
    assert np.array_equal(y, O.spmv(A.rows, row, col, val, x))          # 1.6 M rows, bit for bit
    lhs = ctx.spmv(2.5 * x - 0.75 * z)
    rhs = 2.5 * y - 0.75 * ctx.spmv(z)
    scale = np.abs(A.val).max() * 15 * (np.abs(x).max() + np.abs(z).max())
    assert np.abs(lhs - rhs).max() <= 1e-13 * scale
    d = ctx.dot(x, y)
    assert abs(d - float(np.dot(x, y))) <= 1e-11 * float(np.abs(x * y).sum())


def test_repeated_dropin_calls_cost_a_tenth(big):
    """matrix residency (tests/test_dropin_cache.py) at BASELINE.json's size: the second drop-in call on unchanged
    blocks -- content hash of ~300 MB + the solve itself -- takes a fraction of the first (plan + 250 MB upload): a tenth
    while the plan ran on one host thread, an eighth now that the first call itself is four times cheaper"""
    import ctypes as C
    import os
    import time
    A, _, _ = big
    L = H.lib()
    blk = H.single_rank_blocks(A)
    b = A.matvec(np.ones(A.rows))
    dp = C.POINTER(C.c_double)
    os.environ["BICG_MAX_ITER"] = "4"; os.environ["BICG_QUIET"] = "1"
    try:
        L.bicg_dropin_release()
        t = []
        for _ in range(3):
            x, r = np.zeros(A.rows), b.copy()
            t0 = time.perf_counter()
            L.bicgstab(C.byref(blk.diag), C.byref(blk.offd), C.byref(blk.info), x.ctypes.data_as(dp), r.ctypes.data_as(dp))
            t.append(time.perf_counter() - t0)
        print(f"drop-in call 1 {t[0]:.3f} s, later {min(t[1:]):.3f} s")
        # (round 4: the plan runs on several host threads -- the first call fell from 0.35 to 0.08 s on the GPU box, the later
        # ones stay at 10 ms: a content hash of 300 MB and four iterations)
        assert min(t[1:]) < 0.25 * t[0] and min(t[1:]) < 0.03, t
    finally:
        os.environ.pop("BICG_MAX_ITER", None); os.environ.pop("BICG_QUIET", None)
        L.bicg_dropin_release()


def test_spmm_16_vectors_reads_the_matrix_once(big):
    """BASELINE.json configs[4] "batched SpMV": 16 vectors through one pass over A. Bit-identical columns, and far
    cheaper than 16 SpMVs (the reference's verification loop, src/test_shifted.c:129-154)."""
    A, (row, col, val), ctx = big
    rng = np.random.default_rng(4)
    X = rng.standard_normal((16, A.rows))
    sigma = (np.arange(16) + 1.0) * 0.01 / 16
    Y, ms = ctx.spmm(X, sigma)
    for j in (0, 7, 15):
        assert np.array_equal(Y[j], O.spmv(A.rows, row, col, val, X[j]) + sigma[j] * X[j])
    ctx.spmm(X, sigma)
    ms = min(ctx.spmm(X, sigma)[1] for _ in range(3))          # device time of the pass (windowed kernel: no layout change)
    one = ctx.spmv_bench(100)
    print(f"SpMM 16 vectors {1e3 * ms:.1f} us, one SpMV {1e3 * one:.1f} us")
    # round 4: X staged in LDS per 256-row group straight from the shift-major vectors (k_spmm_win: no layout change, X read
    # once): 345 us against 446 us for the row-major kernel + its transposes -- while one SpMV went from 48 to 35 us.
    # round 6: the pipelined kernel (k_spmm_pipe) ~250 us = 7.5 products of 33 us (the event bracket no longer holds the shifts' upload)
    assert ctx.last_spmm_kind() == "pipelined" and ms <= 0.52 * 16 * one


@pytest.mark.parametrize("method", ["bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"])
def test_first_iterations_and_residual_identity(big, method):
    A, (row, col, val), ctx = big
    b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
    k_fix = 12
    orc = O.solve(method, A.rows, row, col, val, b, tol=0.0, max_iter=k_fix, krr=5, nrr=1)
    got = ctx.solve(method, b, tol=0.0, max_iter=k_fix, krr=5, nrr=1, check_every=k_fix)
    assert got["k"] == orc["k"] == k_fix
    tr = ctx.trace(k_fix)
    for key in ("alpha", "omega", "beta", "dotr"):
        np.testing.assert_allclose(tr[key], orc[key], rtol=1e-7, err_msg=key)
    # the recursive residual the solver returns is the true residual b - A x (to rounding)
    true_r = b - O.spmv(A.rows, row, col, val, got["x"])
    assert np.linalg.norm(true_r - got["r"]) <= 1e-9 * np.linalg.norm(b)
    assert np.abs(got["x"] - orc["x"]).max() <= 1e-8 * np.abs(orc["x"]).max()


def test_hybrid_plan_with_ragged_rows():
    """long / ragged rows push their 256-row groups to the CSR kernel, the rest stays sliced-ELL"""
    H.lib().bicg_comm_init_single(0)
    A = synth.from_offsets(60000, (0, 1, -1, 200, -200), diag_base=8.0, seed=2)
    # make rows 1000..1009 dense-ish (3000 entries each) by rebuilding those rows
    ptr = A.ptr.astype(np.int64)
    cols, vals, lens = [], [], []
    rng = np.random.default_rng(7)
    for r in range(A.rows):
        if 1000 <= r < 1010:
            c = np.sort(rng.choice(A.rows, size=3000, replace=False)); v = rng.uniform(-1e-3, 1e-3, size=3000)
            v[np.searchsorted(c, r) % 3000] += 0.0
        else:
            c, v = A.col[ptr[r]:ptr[r + 1]], A.val[ptr[r]:ptr[r + 1]]
        cols.append(c); vals.append(v); lens.append(len(c))
    p2 = np.zeros(A.rows + 1, dtype=np.int64); np.cumsum(lens, out=p2[1:])
    B = synth.CSR(A.rows, A.rows, p2.astype(np.uint32), np.concatenate(cols).astype(np.uint32), np.concatenate(vals))
    ctx = H.Context(H.single_rank_blocks(B))
    info = ctx.plan_info()
    assert 0 < info["sell_rows"] < B.rows
    x = rng.standard_normal(B.rows)
    row, col, val = B.to_coo()
    y, y_orc = ctx.spmv(x), O.spmv(B.rows, row, col, val, x)
    short = np.diff(p2) <= 2044
    assert np.array_equal(y[short], y_orc[short])
    assert np.abs(y - y_orc).max() <= 1e-12 * np.abs(y_orc).max()
    ctx.close()


def test_jagged_slices_on_fem_like_rows(monkeypatch):
    """ragged rows (synth.fem_like: 27-point stencil with 45 % of the off-diagonals dropped) take the jagged
    sliced-ELL layout -- no padding stored, every row still summed in stored order: SpMV bit-identical to the
    oracle, and identical to what the padded layout / the CSR kernel give; the four solvers follow the oracle."""
    H.lib().bicg_comm_init_single(0)
    A = synth.fem_like(n=117 * 117 * 6)
    row, col, val = A.to_coo()
    rng = np.random.default_rng(5)
    x = rng.standard_normal(A.rows)
    y_orc = O.spmv(A.rows, row, col, val, x)
    ctx = H.Context(H.single_rank_blocks(A))
    fl, info = ctx.flags(), ctx.plan_info()
    assert fl["jagged"] and fl["window"] and fl["all_sell"] and fl["col16"] and info["sell_padding"] == 0
    assert ctx.device_matrix_bytes() <= 10.6 * A.nnz + 8 * A.rows + 4096
    assert np.array_equal(ctx.spmv(x), y_orc)
    b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
    for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"):
        orc = O.solve(method, A.rows, row, col, val, b, krr=10, nrr=2)
        got = ctx.solve(method, b, krr=10, nrr=2) if method.endswith("_rr") else ctx.solve(method, b)
        assert abs(got["k"] - orc["k"]) <= 2, (method, got["k"], orc["k"])
        assert np.abs(got["x"] - orc["x"]).max() <= 1e-8 * np.abs(orc["x"]).max(), method
    ctx.close()
    H.switches(window=0)       # jagged slices, x gathered from memory (16-bit offsets)
    ctx = H.Context(H.single_rank_blocks(A))
    assert ctx.flags()["jagged"] and ctx.flags()["col16"] and not ctx.flags()["window"]
    assert np.array_equal(ctx.spmv(x), y_orc)
    got = ctx.solve("pipe_bicgstab", b)
    orc = O.solve("pipe_bicgstab", A.rows, row, col, val, b)
    assert abs(got["k"] - orc["k"]) <= 2
    ctx.close()
    H.switches(col16=0)          # 32-bit columns in jagged slices
    ctx = H.Context(H.single_rank_blocks(A))
    assert ctx.flags()["jagged"] and not ctx.flags()["col16"]
    assert np.array_equal(ctx.spmv(x), y_orc)
    ctx.close()
    H.switches(col16=None)
    H.switches(window=None)
    H.switches(layout="pad")     # padded slices: most groups fall to the CSR kernel
    ctx = H.Context(H.single_rank_blocks(A))
    assert not ctx.flags()["jagged"]
    assert np.array_equal(ctx.spmv(x), y_orc)
    ctx.close()


def test_ragged_rows_product_with_the_short_chain_gives_the_same_bits(monkeypatch):
    """k_spmv_jagw (csrc/bicg_jagw.hip: window bounds + one 16-bit word per lane, then every run descriptor and the first two
    batches of entries, then the whole window -- three dependent trips per 256-row group) against k_spmv_sell's loop over the same
    jagged slices (BICG_PLAN="jagw=0"): the product, the solvers' scalars (same partial-sum slots, same order) and a shifted solve are
    identical in every bit; the product equals the oracle's mult() (reference src/matrix.c:506-515). Three shapes: FEM-like rows
    (3 runs per group, rows not a multiple of 256), columns in seven separate runs per group, a Transport-sized one."""
    H.lib().bicg_comm_init_single(0)
    monkeypatch.setenv("BICG_PERSIST", "0")
    rng = np.random.default_rng(77)
    n2 = 40000
    scattered = synth.from_offsets(n2, (0, 1, -1, 400, -400, 401, -401, 1500, -1500, 3000, -3000), diag_base=15.0, seed=9)
    # drop entries at random: ragged rows whose columns fall into seven separate runs per 256-row group
    keep = rng.random(scattered.nnz) < 0.6
    keep[scattered.col == np.repeat(np.arange(n2), np.diff(scattered.ptr.astype(np.int64)))] = True
    ptr = np.zeros(n2 + 1, dtype=np.int64)
    np.add.at(ptr, np.repeat(np.arange(n2), np.diff(scattered.ptr.astype(np.int64)))[keep] + 1, 1)
    ragged = synth.CSR(n2, n2, np.cumsum(ptr).astype(np.uint32), scattered.col[keep].copy(), scattered.val[keep].copy())
    # ("fem_short_last_group": 100 rows in the last 256-row group -- its third and fourth slice do not exist, nor does their metadata:
    # a read past the slice arrays once sent a wavefront into a four-billion-step loop whenever that memory was not zero)
    for name, A in (("fem", synth.fem_like(n=117 * 117 * 6)), ("fem_short_last_group", synth.fem_like(n=117 * 117 * 5 + 7)), ("scattered", ragged),
                    ("fem_transport_size", synth.fem_like(scale_decades=2.0))):
        row, col, val = A.to_coo()
        x = rng.standard_normal(A.rows)
        want = O.spmv(A.rows, row, col, val, x)
        b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
        results = []
        for fast in ("0", "1"):
            H.switches(jagw=fast)
            if name == "scattered":
                H.switches(window=1)
            ctx = H.Context(H.single_rank_blocks(A))
            fl = ctx.flags()
            assert fl["jagged"] and fl["window"], (name, fl)
            y = ctx.spmv(x)
            assert np.array_equal(y, want), (name, fast)
            traces = []
            for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab"):
                ctx.solve(method, b, tol=0.0, max_iter=10, check_every=10)
                tr = ctx.trace(10)
                traces.append(np.concatenate([tr[key] for key in ("alpha", "omega", "beta", "dotr")]))
            sh = ctx.solve_shifted(b, np.array([0.0, 0.02, 0.05]), 1, tol=0.0, max_iter=6, check_every=6)
            results.append((y, traces, sh["x"].copy()))
            ctx.close()
            H.switches(window=None)
        (y0, t0, s0), (y1, t1, s1) = results
        assert np.array_equal(y0, y1), name
        for a_, b_ in zip(t0, t1):
            assert np.array_equal(a_, b_), name
        assert np.array_equal(s0, s1), name
    H.switches(jagw=None)


def test_x_window_for_columns_far_from_the_row(monkeypatch):
    """columns further than 32767 from the row (a 3-D stencil's z neighbours at full size): 16-bit offsets do not
    apply, the x window in LDS would -- 16-bit slots, 10 bytes per non-zero, SpMV bit-identical -- but with equal
    rows it is only taken on request (BICG_PLAN="window=1"; measured slower on the 256^3 Laplacian: as many staging
    loads as gathers); a group whose window would not fit LDS switches the whole block back to memory gathers."""
    H.lib().bicg_comm_init_single(0)
    A = synth.from_offsets(150000, (0, 1, -1, 300, -300, 40000, -40000), diag_base=9.0, seed=3)
    row, col, val = A.to_coo()
    x = np.random.default_rng(6).standard_normal(A.rows)
    y_orc = O.spmv(A.rows, row, col, val, x)
    ctx = H.Context(H.single_rank_blocks(A))      # equal rows, perfectly coalesced gathers: padded slices, 32-bit columns
    assert not ctx.flags()["window"] and not ctx.flags()["col16"] and not ctx.flags()["jagged"]
    ctx.close()
    H.switches(window=1)   # on request
    ctx = H.Context(H.single_rank_blocks(A))
    fl = ctx.flags()
    assert fl["window"] and fl["jagged"] and fl["col16"] and fl["all_sell"]
    assert ctx.device_matrix_bytes() <= 10.6 * A.nnz + 8 * A.rows + 4096
    assert np.array_equal(ctx.spmv(x), y_orc)
    b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
    for method in ("bicgstab", "ca_bicgstab"):
        orc = O.solve(method, A.rows, row, col, val, b)
        got = ctx.solve(method, b)
        assert abs(got["k"] - orc["k"]) <= 2 and np.abs(got["x"] - orc["x"]).max() <= 1e-8 * np.abs(orc["x"]).max(), method
    ctx.close()
    H.switches(window=0)
    ctx = H.Context(H.single_rank_blocks(A))
    assert not ctx.flags()["window"] and not ctx.flags()["col16"]
    assert np.array_equal(ctx.spmv(x), y_orc)
    ctx.close()
    H.switches(window=None)
    # 40 scattered columns per row: a 256-row group touches > 4096 distinct x values
    B = synth.random_rows(20000, 40, seed=11)
    ctx = H.Context(H.single_rank_blocks(B))
    assert not ctx.flags()["window"]
    rb, cb, vb = B.to_coo()
    xb = np.random.default_rng(7).standard_normal(B.rows)
    assert np.array_equal(ctx.spmv(xb), O.spmv(B.rows, rb, cb, vb, xb))
    ctx.close()


def test_laplace7_slab_generator_and_ca_bicgstab_against_oracle():
    """BASELINE.json configs[3] family: the in-memory 7-point Laplacian (synth.stencil7 with LAPLACE_WEIGHTS, built
    slab by slab as bench.py does for one GPU's z-planes) at 96^3 = 885 k rows: the slabs tile the global matrix, the
    SpMV is bit-exact against the oracle and CA-BiCGStab follows the oracle's trajectory (src/solver.c:160-278)."""
    H.lib().bicg_comm_init_single(0)
    m = 96
    n = m ** 3
    cuts = [0, 17 * m * m, 48 * m * m, n]                         # three z-slabs of unequal thickness
    slabs = [synth.stencil7(m, synth.LAPLACE_WEIGHTS, rows=(cuts[i], cuts[i + 1])) for i in range(3)]
    A = synth.stencil7(m, synth.LAPLACE_WEIGHTS)
    assert A.nnz == synth.stencil7_nnz(m) == sum(s.nnz for s in slabs)
    assert np.array_equal(np.concatenate([s.col for s in slabs]), A.col) and np.array_equal(np.concatenate([s.val for s in slabs]), A.val)
    row, col, val = A.to_coo()
    ctx = H.Context(H.single_rank_blocks(A))
    assert ctx.plan_info()["sell_rows"] == n
    x = np.random.default_rng(6).standard_normal(n)
    assert np.array_equal(ctx.spmv(x), O.spmv(n, row, col, val, x))
    b = O.spmv(n, row, col, val, np.ones(n))
    for method in ("ca_bicgstab", "bicgstab"):
        orc = O.solve(method, n, row, col, val, b, tol=0.0, max_iter=15)
        got = ctx.solve(method, b, tol=0.0, max_iter=15, check_every=15)
        tr = ctx.trace(15)
        for key in ("alpha", "omega", "beta", "dotr"):
            np.testing.assert_allclose(tr[key], orc[key], rtol=1e-7, err_msg=f"{method} {key}")
        assert np.abs(got["x"] - orc["x"]).max() <= 1e-8 * np.abs(orc["x"]).max()
    ctx.close()


def test_device_side_plan_matches_host_plan():
    """bicg_stencil7_device + bicg_create_device_csr (matrix generated and planned on the GPU, no host copy: what
    bench.py's 512^3 leg uses) against the host path on the same stencil at 96^3: the same plan facts, SpMV bit-identical
    to the oracle (hence to the host-planned context), CA-BiCGStab on the oracle's trajectory; the non-symmetric test
    weights too (every one of the seven positions carries its own value)."""
    H.lib().bicg_comm_init_single(0)
    # m = 192: m^2 = 36 864 > 32 767, the z neighbours need 32-BIT columns -- the branch the 512^3 bench leg takes
    # (the two smaller grids take the packed 16-bit offsets; at 7 M rows the non-symmetric test weights make the CA recurrence so
    # sensitive to the association of the dot sums that alpha moves by 1e-6 within 10 iterations: bench.py's weights there)
    for m, weights in ((96, synth.LAPLACE_WEIGHTS), (33, (6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.0)), (192, synth.LAPLACE_WEIGHTS)):
        A = synth.stencil7(m, weights)
        row, col, val = A.to_coo()
        ctx, nnz, plan_s, gen_s = H.Context.stencil7_on_device(m, weights)
        ref = H.Context(H.single_rank_blocks(A))
        assert nnz == A.nnz == synth.stencil7_nnz(m)
        pi, pr = ctx.plan_info(), ref.plan_info()
        for key in ("rows", "nnz_diag", "sell_rows", "sell_padding", "row_blocks"):
            assert pi[key] == pr[key], (key, pi, pr)
        assert ctx.flags() == ref.flags() or {k: v for k, v in ctx.flags().items() if k != "persist"} == {k: v for k, v in ref.flags().items() if k != "persist"}
        assert ctx.device_matrix_bytes() <= ref.device_matrix_bytes() + 8 * A.rows + 64
        assert ctx.flags()["col16"] == ref.flags()["col16"] == (m * m <= 32767), (m, ctx.flags())
        x = np.random.default_rng(m).standard_normal(A.rows)
        y = ctx.spmv(x)
        assert np.array_equal(y, O.spmv(A.rows, row, col, val, x)) and np.array_equal(y, ref.spmv(x))
        b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
        orc = O.solve("ca_bicgstab", A.rows, row, col, val, b, tol=0.0, max_iter=10)
        got = ctx.solve("ca_bicgstab", b, tol=0.0, max_iter=10, check_every=10)
        tr = ctx.trace(10)
        for key in ("alpha", "omega", "beta", "dotr"):
            np.testing.assert_allclose(tr[key], orc[key], rtol=1e-7, err_msg=key)
        assert np.abs(got["x"] - orc["x"]).max() <= 1e-8 * np.abs(orc["x"]).max()
        print(f"stencil {m}^3: generated on the device in {gen_s:.3f} s, planned in {plan_s:.3f} s")
        ctx.close(); ref.close()


def test_device_plan_survives_hash_collisions(monkeypatch):
    """bicg_create_device_csr groups uniform / constant / masked slices by a 64-bit hash of their lists and fetches ONE
    representative per hash (nothing compared a slice with the list it got). k_plan_verify now does, and a
    slice that differs goes back to its stored columns and values. BICG_TEST=plan-collide throws every hash into one of two
    buckets: thousands of slices get a foreign list, every one has to be caught -- the product stays bit-exact."""
    H.lib().bicg_comm_init_single(0)
    weights = (6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.0)
    for m in (96, 64):
        A = synth.stencil7(m, weights)
        row, col, val = A.to_coo()
        x = np.random.default_rng(m).standard_normal(A.rows)
        want = O.spmv(A.rows, row, col, val, x)
        H.switches(plan_collide=None)
        ctx, _, _, _ = H.Context.stencil7_on_device(m, weights)
        assert ctx.plan_collisions() == 0 and ctx.constant_entries() > 0
        clean_constant = ctx.constant_entries()
        assert np.array_equal(ctx.spmv(x), want)
        ctx.close()
        H.switches(plan_collide=1)
        ctx, _, _, _ = H.Context.stencil7_on_device(m, weights)
        assert ctx.plan_collisions() > 0 and ctx.constant_entries() < clean_constant
        assert np.array_equal(ctx.spmv(x), want)
        b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
        got = ctx.solve("bicgstab", b, tol=1e-10, max_iter=200)
        assert np.abs(got["x"] - 1.0).max() <= 1e-7
        ctx.close()


def test_list_driven_slices_same_bits_whichever_loop_runs(monkeypatch):
    """Constant / masked slices (SellDev::vbase / mbase) have three forms of the product: the general loop of sell_row
    (BICG_PLAN="desc=0"), sell_row with one descriptor per slice and the next slices' metadata requested ahead
    (BICG_PLAN="lists=0"), and the loop of its own for blocks whose slices are ALL list-driven (default; 64^3 and 96^3 qualify,
    33^3 -- rows not a multiple of 256 -- does not). Same sums in the same order (reference src/matrix.c:506-515): SpMV, the
    SpMV with fused dots inside the solvers (alpha / omega / (r,r) of 12 iterations) and the shifted product are bit-identical."""
    H.lib().bicg_comm_init_single(0)
    monkeypatch.setenv("BICG_PERSIST", "0")            # (the one-launch iterations keep the matrix in LDS: not these kernels)
    for m, weights in ((64, synth.LAPLACE_WEIGHTS), (33, (6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.0)), (96, (6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.0))):
        A = synth.stencil7(m, weights)
        x = np.random.default_rng(m).standard_normal(A.rows)
        b = None
        results = []
        # (BICG_PLAN="stencil=0": the 64^3 grid would otherwise go to the plane-marching product, whose dot sums are tiled differently --
        # tests/test_stencil.py compares that one)
        for env in ({"desc": 0}, {"lists": 0}, {"stencil": 0}):
            H.switches(**{k: env.get(k) for k in ("desc", "lists", "stencil")})
            ctx = H.Context(H.single_rank_blocks(A))
            assert ctx.flags()["constant"] and ctx.masked_rows() > 0
            y = ctx.spmv(x)
            if b is None:
                b = ctx.spmv(np.ones(A.rows))
            traces = []
            for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab"):
                ctx.solve(method, b, tol=0.0, max_iter=12, check_every=12)
                tr = ctx.trace(12)
                traces.append(np.concatenate([tr[key] for key in ("alpha", "omega", "beta", "dotr")]))
            sh = ctx.solve_shifted(b, np.array([0.0, 0.02, 0.05]), 1, tol=0.0, max_iter=8, check_every=8)
            results.append((y, traces, sh["x"]))
            ctx.close()
        for y, traces, xs in results[1:]:
            assert np.array_equal(y, results[0][0])
            for t, t0 in zip(traces, results[0][1]):
                assert np.array_equal(t, t0)
            assert np.array_equal(xs, results[0][2])


def test_laplace512_device_plan_at_bench_size():
    """BASELINE.json configs[3] at its stated size, the very calls of bench.py's laplace512(): 134 M rows / 938 M non-zeros
    generated and planned on the GPU with 32-bit columns. No CPU oracle run fits a test at this size, so the SpMV is compared
    with the CLOSED FORM of the stencil (synth.stencil7_matvec: stored-order sums without FMA, bit-identical to the oracle's
    mult() -- tests/test_host_logic.py) -- every one of the seven positions with its own weight -- and the solvers through
    size-independent properties: b = A 1 vanishes in the interior, after 6 iterations of CA-BiCGStab and plain BiCGStab the
    TRUE residual b - A x (closed form) equals the recursive one, and the (r,r) the device reports is that of the r it
    returns (reference src/solver.c:160-278, 35-146; src/matrix.c:498-516)."""
    H.lib().bicg_comm_init_single(0)
    m = 512
    n = m ** 3
    w = (6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.0)
    ctx, nnz, plan_s, gen_s = H.Context.stencil7_on_device(m, w)
    assert nnz == synth.stencil7_nnz(m) and ctx.plan_info()["sell_rows"] == n and not ctx.flags()["col16"]
    st = ctx.stencil_info()          # one value per distance, vectors beyond the Infinity Cache: two rows per lane, 128 x 16 x 64 tiles
    assert st["on"] == 1 and st["rows_per_lane"] == 2 and st["lines"] == 4 and st["planes"] == 64 and st["workgroups"] == 1024, st
    x = np.random.default_rng(512).standard_normal(n)
    assert np.array_equal(ctx.spmv(x), synth.stencil7_matvec(m, w, x))           # 134 M rows, bit for bit
    ctx.close()
    ctx, nnz, plan_s, gen_s = H.Context.stencil7_on_device(m, synth.LAPLACE_WEIGHTS)
    ones = np.ones(n)
    b = ctx.spmv(ones)
    assert np.array_equal(b, synth.stencil7_matvec(m, synth.LAPLACE_WEIGHTS, ones))
    assert np.all(b.reshape(m, m, m)[1:-1, 1:-1, 1:-1] == 0.0) and b.reshape(m, m, m)[0, 0, 0] == 3.0
    nb = float(np.sqrt(b @ b))
    for method in ("ca_bicgstab", "bicgstab"):
        got = ctx.solve(method, b, tol=0.0, max_iter=6, check_every=6)
        assert got["k"] == 6 and got["result"].breakdown_iteration == 0
        true_r = b - synth.stencil7_matvec(m, synth.LAPLACE_WEIGHTS, got["x"])
        assert float(np.sqrt(((true_r - got["r"]) ** 2).sum())) <= 1e-12 * nb, method
        rr = float(got["r"] @ got["r"])
        assert abs(got["result"].dot_r - rr) <= 1e-12 * rr and 0.0 < rr < float(b @ b), method
    print(f"stencil 512^3: generated on the device in {gen_s:.3f} s, planned in {plan_s:.3f} s")
    ctx.close()
