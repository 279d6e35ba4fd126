"""The multi-rank machinery on ONE GPU (`-m gpu`): BICG_TEST=force-comm makes a single rank take the
N>1 code path -- halo pack/exchange calls, packed all-reduce of every dot group on the
communication stream, 1-thread apply kernels, event joins -- with either the trivial transport or a
REAL one-rank RCCL communicator, eagerly and as a replayed hipGraph. Every combination must
reproduce the plain single-rank result bit for bit (the arithmetic is the same; only where the
scalar recurrences run changes)."""
import ctypes as C

import numpy as np
import pytest

from mpi_bicgstab_amd import hipsolver as H
from mpi_bicgstab_amd import synth

pytestmark = pytest.mark.gpu

METHODS = [("bicgstab", 1e-15), ("ca_bicgstab", 1e-15), ("pipe_bicgstab", 1e-9), ("pipe_bicgstab_rr", 1e-15)]
SHIFTED = ["shifted_lopbicgstab", "shifted_pipe_lopbicgstab", "shifted_bicgstab"]


def _solve_all(A, b, **env):
    import os
    # bit equality across transports is a statement about identical kernels: the persistent one-launch form of
    # pipe_bicgstab (bicg_persist.hip) associates the dot sums differently and is compared on its own below
    # (lower-case keywords are tokens of BICG_PLAN / BICG_PERSIST / BICG_TEST -- hipsolver.switches --, the rest are variables)
    env.setdefault("persist", 0)
    saved = {k: os.environ.get(k) for k in list(H.SWITCH_VARS) + [k for k in env if k not in H.SWITCHES]}
    H.switches(**{k: v for k, v in env.items() if k in H.SWITCHES})
    os.environ.update({k: str(v) for k, v in env.items() if k not in H.SWITCHES})
    try:
        ctx = H.Context(H.single_rank_blocks(A))
        out = {}
        for m, tol in METHODS:
            r = ctx.solve(m, b, tol=tol, krr=10, nrr=3, check_every=5)
            out[m] = (r["k"], r["x"].copy(), r["r"].copy())
        sigma = 0.01 * (np.arange(4) + 1.0)
        for which in SHIFTED:      # b doubles as the right-hand side of the seed system here
            r = ctx.solve_shifted(b, sigma, 1, which=which, tol=1e-11, check_every=5)
            out[which] = (r["k"], r["x"].copy(), r["r"].copy())
        ctx.close()
        return out
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def problem():
    H.lib().bicg_comm_init_single(0)
    A = synth.from_offsets(20011, (0, 1, -1, 60, -60, 61, -61, 3000, -3000), diag_base=11.0, seed=4)
    ctx = H.Context(H.single_rank_blocks(A))
    b = ctx.spmv(np.ones(A.rows))
    ctx.close()
    # BICG_PLAN="fuse-pipe=0": a small single rank runs the pipelined phases in the SpMV epilogues (two launches per iteration),
    # the host-enqueued transports keep them as separate kernels; same arithmetic per element, but the dot sums are then
    # associated differently -- bit equality across transports is a statement about identical kernels
    return A, b, _solve_all(A, b, BICG_GRAPH=0, fuse_pipe=0)


def _same(got, ref):
    for m in [m for m, _ in METHODS] + SHIFTED:
        assert got[m][0] == ref[m][0], m
        assert np.array_equal(got[m][1], ref[m][1]) and np.array_equal(got[m][2], ref[m][2]), m


def test_graph_replay_single_rank(problem):
    A, b, ref = problem
    _same(_solve_all(A, b, BICG_GRAPH=1, fuse_pipe=0), ref)


@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("graph", [0, 1])
def test_forced_comm_trivial_transport(problem, graph, overlap):
    A, b, ref = problem
    H.lib().bicg_comm_init_single(0)
    _same(_solve_all(A, b, force_comm=1, BICG_GRAPH=graph, BICG_OVERLAP=overlap), ref)


@pytest.mark.parametrize("overlap", [0, 1])      # 1 = two-stream mode: halo / all-reduce on the communication stream
@pytest.mark.parametrize("graph", [0, 1])
def test_forced_comm_one_rank_rccl(problem, graph, overlap, monkeypatch):
    A, b, ref = problem
    H.switches(force_comm=1)
    buf = (C.c_char * 128)()
    H.lib().bicg_comm_unique_id(buf)
    H.lib().bicg_comm_init_rccl(0, 1, buf.raw, 0)
    try:
        _same(_solve_all(A, b, force_comm=1, BICG_GRAPH=graph, BICG_OVERLAP=overlap), ref)
    finally:
        H.switches(force_comm=None)
        H.lib().bicg_comm_init_single(0)


def test_forced_comm_peer_to_peer_one_rank(problem, monkeypatch):
    """The peer-to-peer data path (bicg_p2p.cpp) with one rank: dot groups go through the LL mailbox
    and k_apply_p2p, SpMVs through the (empty) push / unpack kernels -- same bits as single rank."""
    A, b, ref = problem
    H.switches(force_comm=1)
    buf = (C.c_char * 128)()
    H.lib().bicg_comm_unique_id(buf)
    H.lib().bicg_comm_init_rccl(0, 1, buf.raw, 0)
    try:
        assert H.lib().bicg_comm_enable_p2p() == 0
        assert H.lib().bicg_comm_p2p_active() > 0
        _same(_solve_all(A, b, force_comm=1, BICG_GRAPH=0, fuse_pipe=0), ref)
    finally:
        H.switches(force_comm=None)
        H.lib().bicg_comm_init_single(0)


def test_fused_pipelined_iteration_matches_separate_kernels(problem):
    """k_spmv_sell_epi (phases in the SpMV epilogues) against the four-kernel iteration: same iteration count, same
    solution to rounding (only the association of the dot sums differs)"""
    A, b, ref = problem
    got = _solve_all(A, b, BICG_GRAPH=0, fuse_pipe=1)
    for m in ("pipe_bicgstab", "pipe_bicgstab_rr"):
        assert abs(got[m][0] - ref[m][0]) <= 1, m
        assert np.abs(got[m][1] - ref[m][1]).max() <= 1e-9 * np.abs(ref[m][1]).max(), m
    for m in ("bicgstab", "ca_bicgstab"):          # untouched by the switch
        assert got[m][0] == ref[m][0] and np.array_equal(got[m][1], ref[m][1])


def test_persistent_iterations(problem, monkeypatch):
    """pipe_bicgstab, bicgstab and ca_bicgstab as ONE persistent launch per chunk of iterations (bicg_persist.hip: matrix
    slices and x window in LDS, vectors in registers, everything that crosses workgroups as LL words) against the
    multi-launch forms: same iteration count +-1, same solution to rounding (the dot sums are associated per wavefront ->
    workgroup -> table); bit-reproducible from run to run; independent of how often the host looks (chunk length); and the
    same bits through the peer-to-peer transport driven by one rank (helper workgroup exchanging the sums through the
    mailboxes). pipe_bicgstab_rr (replacement steps) keeps the multi-launch form."""
    A, b, ref = problem
    H.lib().bicg_comm_init_single(0)
    tols = dict(METHODS)
    methods = ("pipe_bicgstab", "bicgstab", "ca_bicgstab")
    runs = []
    # (BICG_PERSIST="chunk=n": a persistent launch covers at least that many iterations whatever check_every says -- 128 by
    # default; 1 = exactly check_every, the last run takes the default)
    for check_every, least in ((5, 1), (5, 1), (1, 1), (64, 1), (5, None)):
        H.switches(persist=1, persist_chunk=least)
        ctx = H.Context(H.single_rank_blocks(A))
        assert ctx.flags()["persist"], ctx.flags()
        out = {}
        for m in methods:
            r = ctx.solve(m, b, tol=tols[m], check_every=check_every)
            tr = ctx.trace(r["k"])
            out[m] = (r["k"], r["x"].copy(), r["r"].copy(), tr["alpha"].copy(), tr["dotr"].copy())
        r = ctx.solve("pipe_bicgstab_rr", b, tol=1e-15, krr=10, nrr=3, check_every=5)
        # (the replacement variant keeps its multi-launch forms; the default two-launch one against the separate kernels of ref)
        assert abs(r["k"] - ref["pipe_bicgstab_rr"][0]) <= 1 and np.abs(r["x"] - ref["pipe_bicgstab_rr"][1]).max() <= 1e-9
        runs.append(out)
        ctx.close()
    for m in methods:
        k0, x0, r0, a0, d0 = runs[0][m]
        assert abs(k0 - ref[m][0]) <= 1, m
        assert np.abs(x0 - ref[m][1]).max() <= 1e-9 * np.abs(ref[m][1]).max(), m
        for other in runs[1:]:
            k, x, r, a, d = other[m]
            assert k == k0 and np.array_equal(x, x0) and np.array_equal(r, r0) and np.array_equal(a, a0) and np.array_equal(d, d0), m
    # through the peer-to-peer transport, one rank: dot groups via the mailboxes, (empty) halo pushes
    H.switches(force_comm=1)
    buf = (C.c_char * 128)()
    H.lib().bicg_comm_unique_id(buf)
    H.lib().bicg_comm_init_rccl(0, 1, buf.raw, 0)
    try:
        assert H.lib().bicg_comm_enable_p2p() == 0
        ctx = H.Context(H.single_rank_blocks(A))
        assert ctx.flags()["persist"] and ctx.flags()["p2p"], ctx.flags()
        for m in methods:
            r = ctx.solve(m, b, tol=tols[m], check_every=5)
            k0, x0, r0, _, _ = runs[0][m]
            assert r["k"] == k0 and np.array_equal(r["x"], x0) and np.array_equal(r["r"], r0), m
        ctx.close()
    finally:
        H.switches(force_comm=None)
        H.lib().bicg_comm_init_single(0)


@pytest.mark.parametrize("method", ["bicgstab", "ca_bicgstab"])
def test_persistent_plain_two_rows_per_thread(method):
    """A rank of the headline matrix at 4 GPUs (400 528 rows of the Transport-shaped matrix: 1 565 rows per CU) takes plain BiCGStab
    and CA-BiCGStab as ONE persistent launch per chunk with two rows per thread (k_plain_persist_r / k_ca_persist_r, round 6;
    reference src/solver.c:86-121, 217-251):
    no product kernel is launched during the solve, the first 12 iterations follow the oracle's scalars, the result equals the
    five-launch iteration's to rounding, run to run and chunk length to chunk length in bits, and the same bits come through the
    peer-to-peer transport driven by one rank (the MULTI instantiation: sums through the mailboxes)."""
    import oracle_lib as O
    H.lib().bicg_comm_init_single(0)
    n = (synth.TRANSPORT_N + 3) // 4
    A = synth.transport_like(n=n, scale_decades=2.0)
    row, col, val = A.to_coo()
    b = O.spmv(n, row, col, val, np.ones(n))
    orc = O.solve(method, n, row, col, val, b, tol=0.0, max_iter=12)
    want = np.concatenate([orc[key][:12] for key in ("alpha", "omega", "beta", "dotr")])

    def run(check_every, **sw):
        H.switches(**sw)
        try:
            ctx = H.Context(H.single_rank_blocks(A))
            ctx.load(np.zeros(n), b)
            ctx.run_begin(method, tol=0.0, max_iter=12, check_every=check_every)      # (r = b - A x0 [, w = A r]: product kernels)
            ctx.sync()
            H.product_kernels()
            ctx.run_iterate(12)
            ctx.sync()
            ran = H.product_kernels()
            assert ctx.run_end().iterations == 12
            x, r = ctx.fetch()
            tr = ctx.trace(12)
            out = (np.concatenate([tr[key] for key in ("alpha", "omega", "beta", "dotr")]), x, r, ran, ctx.flags())
            ctx.close()
            return out
        finally:
            H.switches(**{k: None for k in sw})

    multi = run(12, persist=0)
    assert multi[3] and not multi[4]["persist"], (multi[3], multi[4])
    one = run(12, persist_chunk=1)
    assert one[4]["persist"] and not one[3], ("a product kernel ran: the persistent form was not taken", one[3], one[4])
    np.testing.assert_allclose(one[0], want, rtol=1e-8, atol=0.0)
    np.testing.assert_allclose(one[0], multi[0], rtol=1e-8, atol=0.0)
    np.testing.assert_allclose(one[1], multi[1], rtol=1e-8, atol=1e-12)
    for other in (run(12, persist_chunk=1), run(4, persist_chunk=1), run(1, persist_chunk=1)):
        assert not other[3]
        assert np.array_equal(other[0], one[0]) and np.array_equal(other[1], one[1]) and np.array_equal(other[2], one[2])
    H.switches(force_comm=1)
    buf = (C.c_char * 128)()
    H.lib().bicg_comm_unique_id(buf)
    H.lib().bicg_comm_init_rccl(0, 1, buf.raw, 0)
    try:
        assert H.lib().bicg_comm_enable_p2p() == 0
        p2p = run(6, persist_chunk=1)
        assert p2p[4]["persist"] and p2p[4]["p2p"] and not p2p[3], (p2p[3], p2p[4])
        assert np.array_equal(p2p[0], one[0]) and np.array_equal(p2p[1], one[1]) and np.array_equal(p2p[2], one[2])
    finally:
        H.switches(force_comm=None)
        H.lib().bicg_comm_init_single(0)


def test_fused_pipelined_iteration_is_bit_reproducible(problem, monkeypatch):
    """the two-launch form (k_spmv_sell_epi) twice, and once more with BICG_TEST="spin-ticks=0" -- every workgroup then sums
    the shards it is waiting for itself (same partials, same order): same bits whoever computes a shard"""
    A, b, ref = problem
    H.lib().bicg_comm_init_single(0)
    a1 = _solve_all(A, b, BICG_GRAPH=0, fuse_pipe=1)
    a2 = _solve_all(A, b, BICG_GRAPH=0, fuse_pipe=1)
    a3 = _solve_all(A, b, BICG_GRAPH=0, fuse_pipe=1, spin_ticks=0)
    for m in ("pipe_bicgstab", "pipe_bicgstab_rr"):
        for other in (a2, a3):
            assert other[m][0] == a1[m][0] and np.array_equal(other[m][1], a1[m][1]) and np.array_equal(other[m][2], a1[m][2]), m


def test_process_with_rccl_and_torch_exits_cleanly():
    """A real one-rank RCCL round trip through the library, THEN `import torch`, then a normal exit: the process
    status must be 0 (see tests/test_host_logic.py::test_rccl_loaded_by_the_library_then_torch_exits_cleanly -- with
    librccl's dependencies in the global symbol scope, the pytest run over this module and any torch-importing one
    ended in exit status 134 with every test green)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mpi_bicgstab_amd import hipsolver as H\n"
            "assert H.lib().bicg_comm_selftest_rccl(0) == 0\n"
            "import torch\nassert torch.cuda.is_available()\nprint('rccl then torch')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl then torch" in out.stdout, (out.returncode, out.stderr[-400:])


@pytest.mark.parametrize("overlap", [0, 1])
def test_section_times_on_the_multi_rank_path(problem, overlap, monkeypatch):
    """Section timing (bicg_options.time_kernels & 2) on the N > 1 code path: the all-reduce + apply hand-overs are
    launches of their own there and must show up as the reduction section (the reference's `ared` column,
    src/shifted_switching_solver.c:884-892); the marks must not change a bit of the result."""
    A, b, ref = problem
    H.lib().bicg_comm_init_single(0)
    H.switches(force_comm=1, persist=0)
    for k, v in dict(BICG_GRAPH=0, BICG_OVERLAP=overlap).items():
        monkeypatch.setenv(k, str(v))
    ctx = H.Context(H.single_rank_blocks(A))
    for m, tol in METHODS:
        r = ctx.solve(m, b, tol=tol, krr=10, nrr=3, check_every=5, time_kernels=2)
        assert r["k"] == ref[m][0] and np.array_equal(r["x"], ref[m][1]) and np.array_equal(r["r"], ref[m][2]), m
        t = ctx.section_times()
        assert t is not None and t["iterations"] == r["k"], (m, t)
        assert t["spmv_ms"] > 0.0 and t["vec_ms"] > 0.0 and t["shift_ms"] == 0.0, (m, t)
        # two-stream mode hides the pipelined solvers' all-reduces behind the next product (src/solver.c:363-367): they are
        # started and joined inside spmv() and counted there
        if not (overlap and m.startswith("pipe")):
            assert t["reduce_ms"] > 0.0, (m, t)
    ctx.close()


def test_pipelined_form_chosen_by_measurement(monkeypatch):
    """BICG_PLAN=pipe-probe: the first pipelined solve on a context times both multi-launch forms (phases as separate kernels /
    in the SpMV epilogues) on the caller's own system and keeps the faster one -- instead of the size / layout rule of
    bicg_create. The probe must leave x0 and b untouched: the solve that follows is bit for bit the solve of a context
    pinned to the chosen form (BICG_PLAN="fuse-pipe=0|1"), also for a second call and for the replacement variant."""
    H.lib().bicg_comm_init_single(0)
    A = synth.fem_like(300_000, scale_decades=1.0)
    H.switches(persist=0, fuse_pipe=None, pipe_probe=1)
    ctx = H.Context(H.single_rank_blocks(A))
    b = ctx.spmv(np.ones(A.rows))
    x0 = np.random.default_rng(2).standard_normal(A.rows) * 1e-3
    assert not ctx.flags()["pipe_probed"]
    got = ctx.solve("pipe_bicgstab", b, x0=x0, tol=0.0, max_iter=40, check_every=8)
    fl = ctx.flags()
    assert fl["pipe_probed"], fl
    again = ctx.solve("pipe_bicgstab", b, x0=x0, tol=0.0, max_iter=40, check_every=8)
    rr = ctx.solve("pipe_bicgstab_rr", b, x0=x0, tol=0.0, max_iter=40, check_every=8, krr=10, nrr=2)
    assert ctx.flags()["fuse_pipe"] == fl["fuse_pipe"]          # decided once per context
    ctx.close()
    H.switches(pipe_probe=None, fuse_pipe=1 if fl["fuse_pipe"] else 0)
    pinned = H.Context(H.single_rank_blocks(A))
    assert not pinned.flags()["pipe_probed"] and pinned.flags()["fuse_pipe"] == fl["fuse_pipe"]
    want = pinned.solve("pipe_bicgstab", b, x0=x0, tol=0.0, max_iter=40, check_every=8)
    want_rr = pinned.solve("pipe_bicgstab_rr", b, x0=x0, tol=0.0, max_iter=40, check_every=8, krr=10, nrr=2)
    pinned.close()
    for a, w in ((got, want), (again, want), (rr, want_rr)):
        assert a["k"] == w["k"] == 40 and np.array_equal(a["x"], w["x"]) and np.array_equal(a["r"], w["r"])
