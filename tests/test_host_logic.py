"""CPU-only checks of the C ABI library: it loads without a GPU, exports every symbol that
include/bicgstab_hip.h declares, keeps the reference's struct layouts, and its host-side planning
(partition, halo plan, row blocks) agrees with the oracle / with first principles."""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from mpi_bicgstab_amd import hipsolver as H
from mpi_bicgstab_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "bicgstab_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bicg_[a-z0-9_]+|shifted_[a-z0-9_]+|bicgstab|ca_bicgstab|pipe_bicgstab|pipe_bicgstab_rr)\s*\(", hdr))
    declared -= {"bicg_allreduce_fn", "bicg_alltoallv_fn"}
    lib = H.lib()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(H.EXPORTS) <= declared, sorted(set(H.EXPORTS) - declared)
    assert len(declared) >= 50


def test_struct_layouts_match_reference():
    # reference src/matrix.h:19-33 on LP64: 3 pointers + 3 u32 (+pad) = 40 ; 3 u32 + char[4] + 2 pointers = 32
    assert C.sizeof(H.CSRMatrix) == 40 and C.sizeof(H.InfoMatrix) == 32
    assert H.CSRMatrix.nz.offset == 24 and H.CSRMatrix.cols.offset == 32
    assert H.InfoMatrix.code.offset == 12 and H.InfoMatrix.recvcounts.offset == 16 and H.InfoMatrix.displs.offset == 24


@pytest.mark.parametrize("n,p", [(10, 4), (1602111, 8), (7, 7), (5, 2), (100, 1)])
def test_partition_matches_reference_rule(n, p):
    cnt, dsp = H.partition(n, p)
    base, extra = divmod(n, p)       # reference src/matrix.c:295-298
    assert list(cnt) == [base + (1 if r < extra else 0) for r in range(p)]
    assert list(dsp) == [r * base + min(r, extra) for r in range(p)]
    c2, d2 = synth.partition(n, p)
    assert np.array_equal(cnt, c2) and np.array_equal(dsp, d2)


def test_transport_partition_sizes():
    cnt, _ = H.partition(1602111, 8)   # 200 264 x 7 + 200 263
    assert list(cnt) == [200264] * 7 + [200263]


@pytest.mark.parametrize("world", [2, 3, 5])
def test_halo_plan(world):
    A = synth.from_offsets(3001, (0, 2, -2, 40, -40, 900, -900), diag_base=9.0)
    for rank in range(world):
        diag, offd, counts, displs = synth.split_blocks(A, world, rank)
        blk = H.HostBlocks(diag, offd, A.rows, counts, displs)
        h, cols, rc, ren = H.halo_plan(blk, world)
        uniq = np.unique(offd.col)
        assert h == len(uniq) and np.array_equal(cols, uniq)
        owner = np.searchsorted(np.asarray(displs), uniq, side="right") - 1
        assert np.array_equal(rc, np.bincount(owner, minlength=world))
        assert rc[rank] == 0
        assert np.array_equal(cols[ren - blk.n_loc], offd.col)      # renumbering is consistent
        assert sum(rc) == h


def test_row_blocks_cover_and_respect_chunk():
    A = synth.random_rows(5000, 60, seed=2, empty_frac=0.2, long_rows={100: 3000, 4999: 2500})
    rb = H.row_blocks(A.ptr, chunk=2048, max_rows=1024)
    assert rb[0] == 0 and rb[-1] == A.rows and np.all(np.diff(rb.astype(np.int64)) >= 1)
    ptr = A.ptr.astype(np.int64)
    nnz = ptr[rb[1:]] - ptr[rb[:-1]]
    rows = np.diff(rb.astype(np.int64))
    assert np.all((nnz <= 2048) | (rows == 1))      # only single (long) rows may exceed the chunk
    assert np.all(rows <= 1024)
    # greedy: a block could not have taken the next row as well
    for i in range(len(rb) - 2):
        nxt = ptr[rb[i + 1] + 1] - ptr[rb[i]]
        assert nxt > 2048 or rows[i] == 1024 or (rows[i] == 1 and nnz[i] > 2048)


def test_no_cpu_fallback_exists():
    """the product path has no route around the HIP library: creating a context without a GPU dies
    loudly (exit from the C library), it does not compute on the CPU."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from mpi_bicgstab_amd import hipsolver as H, synth\n"
            "A = synth.stencil7(4)\n"
            "ctx = H.Context(H.single_rank_blocks(A))\n"
            "print('computed', ctx.spmv(np.ones(A.rows)).sum())\n" % ROOT)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "computed" not in out.stdout
    assert "bicgstab_hip" in out.stderr or "hip" in out.stderr.lower()


def test_synth_matches_oracle_partition():
    A = synth.stencil7(7)
    row, col, val = A.to_coo()
    x = np.random.default_rng(0).standard_normal(A.rows)
    y1 = O.spmv(A.rows, row, col, val, x, nranks=1)
    y3 = O.spmv(A.rows, row, col, val, x, nranks=3)
    assert np.abs(y1 - y3).max() <= 1e-13 * np.abs(y1).max()
    assert np.abs(A.matvec(x) - y1).max() <= 1e-13 * np.abs(y1).max()


def test_header_is_plain_c_and_coexists_with_reference_headers(tmp_path):
    """include/bicgstab_hip.h compiles as C99 on its own, and after the reference's solver.h (whose
    matrix.h typedefs it then reuses) when /root/reference is mounted"""
    import subprocess
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "a.c"
    src.write_text('#include "bicgstab_hip.h"\n'
                   'typedef int (*solver_fn)(CSR_Matrix *, CSR_Matrix *, INFO_Matrix *, double *, double *);\n'
                   'solver_fn pick(int i) { bicg_options o; bicg_default_options(&o); return i ? bicgstab : pipe_bicgstab; }\n'
                   'int sizes(void) { return (int)sizeof(CSR_Matrix) + (int)sizeof(INFO_Matrix); }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", f"-I{inc}", str(src)], check=True)
    ref = "/root/reference/src"
    if os.path.exists(os.path.join(ref, "solver.h")) and os.path.exists("/opt/conda/include/mpi.h"):
        src2 = tmp_path / "b.c"
        src2.write_text('#include "solver.h"\n#include "bicgstab_hip.h"\n'
                        'int g(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r) { return bicgstab(d, o, i, x, r); }\n')
        subprocess.run(["gcc", "-std=gnu99", "-w", "-fsyntax-only", f"-I{ref}", "-I/opt/conda/include", f"-I{inc}", str(src2)],
                       check=True)


def test_partition_nnz_edge_cases():
    """bicg_partition_nnz: contiguous, covers every row once, >= 1 row per rank when n >= P, and no
    block further than one (longest) row from the ideal share."""
    import ctypes as C
    from mpi_bicgstab_amd import hipsolver as H
    L = H.lib()
    ip, upp = C.POINTER(C.c_int), C.POINTER(C.c_uint)

    def cut(lens, P):
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        counts, displs = np.zeros(P, dtype=np.int32), np.zeros(P, dtype=np.int32)
        L.bicg_partition_nnz(lens.ctypes.data_as(upp), C.c_uint(len(lens)), C.c_int(P), counts.ctypes.data_as(ip),
                             displs.ctypes.data_as(ip))
        assert displs[0] == 0 and counts.sum() == len(lens) and (counts >= 0).all()
        assert np.array_equal(displs[1:], np.cumsum(counts)[:-1])
        return counts, displs

    rng = np.random.default_rng(0)
    for n, P in ((1000, 7), (64, 8), (5, 5), (3, 8), (1, 1), (200, 3)):
        lens = rng.integers(0, 40, size=n)
        counts, displs = cut(lens, P)
        if n >= P:
            assert (counts >= 1).all()
            share = np.array([lens[displs[p]:displs[p] + counts[p]].sum() for p in range(P)])
            assert np.abs(share - lens.sum() / P).max() <= max(lens.max(), 1) * (2 if n < 4 * P else 1) + 1
    counts, _ = cut(np.zeros(50), 4)                       # no non-zeros at all
    assert (counts >= 1).all()
    lens = np.ones(100); lens[10] = 10000                  # one dominant row
    counts, displs = cut(lens, 4)
    assert (counts >= 1).all()
    uniform, _ = cut(np.full(1000, 15), 8)                 # uniform rows: the equal-rows partition
    assert np.array_equal(uniform, np.full(8, 125))


def test_hot_kernels_stay_lean():
    """The compiler's resource report of the last build: the sliced-ELL SpMV on the single-GPU hot path (no
    offd, no in-kernel halo exchange; ticket-light and consumer-side-finish instantiations) must stay at
    <= 64 VGPRs, occupancy 8, no scratch -- rarely used paths (seed switching, peer-to-peer collect) once
    leaked into it and cost 26 VGPRs and three waves per SIMD. The element-wise
    kernels of the four solvers hold the loads of their first element pair across the wait for the dot
    sums: more registers by design, but never scratch and at least 5 waves per SIMD."""
    path = os.path.join(ROOT, "mpi-bicgstab_amd", "build", "kernel_resources.txt")
    if not os.path.exists(path):
        pytest.skip("no resource report (library not built here)")
    kernels, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+):\s*(\d+)", line)
        if m and cur:
            kernels[cur][m.group(1).strip()] = int(m.group(2))
    spmv = [k for k in kernels if re.search(r"k_spmv_sellILi[0-3]ELb0ELb[01]ELi[0-3]ELb0ELi[02]EEEv", k)]   # no offd, no LL
    assert len(spmv) >= 48, sorted(kernels)[:5]
    for k in spmv:
        r = kernels[k]
        # ticket mode is pinned to 8 waves per SIMD; since round 4 (slots per row group, XCD-contiguous and alternating order,
        # uniform and constant slices: more kernel arguments live across the launch) the compiler parks up to 11 registers per lane in scratch
        # (calling the hand-over out of line instead costs a 1 KB stack frame: tried).
        # Checked in the disassembly: one store before the group loop, one reload per 256-row group, nothing inside the
        # loop over a row's entries (profiles/NOTES.md, round 4)
        ticket = k.endswith("ELi0EEEvNS_8SpmvArgsE")
        assert r["VGPRs"] <= (64 if ticket else 80) and r["Occupancy [waves/SIMD]"] >= (8 if ticket else 6), (k, r)
        assert r["ScratchSize [bytes/lane]"] <= (48 if ticket else 0), (k, r)
    names = r"k_vecINS_(8FPlainXRILb[01]E|7FPlainQ|7FPlainP|6FPipe1|6FPipe2ILb[01]E|5FCaXRILb[01]E|3FQY|5FCaPS)E*"
    vec = [k for k in kernels if re.search(names + r"ELi2ELi0EEEv", k)]
    assert len(vec) >= 11, sorted(kernels)[:5]
    for k in vec:
        r = kernels[k]
        assert r["VGPRs"] <= 96 and r["Occupancy [waves/SIMD]"] >= 5 and r["ScratchSize [bytes/lane]"] == 0, (k, r)
    # round 5: the tiled form for vectors far beyond the caches (k_vec<.., TILE = 4>: the loads of four element pairs per thread in
    # flight before the first store): registers by design -- 16 streams' worth of them for the pipelined phases --, never scratch
    tiled = [k for k in kernels if re.search(names + r"ELi[0-2]ELi4EEEv", k)]
    assert len(tiled) >= 20, sorted(kernels)[:5]
    for k in tiled:
        r = kernels[k]
        assert r["ScratchSize [bytes/lane]"] == 0 and r["Occupancy [waves/SIMD]"] >= 2, (k, r)
    # ... and the plane-marching product (bicg_stencil.hip): four sets of plane registers, never scratch, four wavefronts per SIMD
    # with four lines per wavefront
    st = [k for k in kernels if "k_spmv_stencilI" in k]
    assert len(st) >= 28
    for k in st:
        assert kernels[k]["ScratchSize [bytes/lane]"] == 0 and kernels[k]["Occupancy [waves/SIMD]"] >= 4, (k, kernels[k])
    # ... its wide form (2 / 4 rows per lane: the plane registers are 16 / 32 bytes each): never scratch
    stw = [k for k in kernels if "k_spmv_stencil_wI" in k]
    assert len(stw) >= 28
    for k in stw:
        assert kernels[k]["ScratchSize [bytes/lane]"] == 0, (k, kernels[k])
    # the fused pipelined iteration: a 200k-row rank is 783 workgroups x 4 wavefronts = 3.06 per SIMD, so a
    # fifth VGPR over 128 (occupancy 3) buys a second round of workgroups: +5 us per iteration, measured
    epi = [k for k in kernels if "k_spmv_sell_epi" in k]
    assert len(epi) >= 16
    for k in epi:
        assert kernels[k]["VGPRs"] <= 128 and kernels[k]["Occupancy [waves/SIMD]"] >= 4, (k, kernels[k])
    # round 3. The persistent kernels: up to 16 wavefronts of one workgroup per CU = 4 per SIMD = 128 VGPRs each; the
    # rows-over-lanes SpMV and the window-fused SpMV at full occupancy
    # round 4: the pipelined kernel with one or two rows per thread (16 wavefronts per CU, 128 registers) and with eight (8
    # wavefronts per CU = 2 per SIMD, 256 registers)
    # round 6: plain and CA-BiCGStab with two rows per thread (k_plain_persist_r / k_ca_persist_r: ranks of 400 k rows), same budget
    persist = [k for k in kernels if re.search(r"k_(pipe|plain|ca)_persist", k)]
    assert len(persist) == 20 and len([k for k in persist if "_persist_rILi2E" in k]) == 4, persist
    shifted = [k for k in kernels if re.search(r"k_sh(pipe|lop)_persist", k)]      # the shifted solvers' forms: same budget
    assert len(shifted) == 8 and all(kernels[k]["VGPRs"] <= 128 and kernels[k]["ScratchSize [bytes/lane]"] <= 32 for k in shifted), shifted      # (round 5: the launch arguments grew by the wait log's pointer: one more parked register in one of them)
    for k in persist:
        if "k_pipe_persistILi8E" in k:
            assert kernels[k]["Occupancy [waves/SIMD]"] >= 2 and kernels[k]["ScratchSize [bytes/lane]"] <= 384, (k, kernels[k])
        else:
            assert kernels[k]["VGPRs"] <= 128 and kernels[k]["Occupancy [waves/SIMD]"] >= 4, (k, kernels[k])
    rows = [k for k in kernels if re.search(r"k_spmv_rowsILi[0-3]ELb0ELb[01]ELb[01]ELi0EEEv", k)]      # no offd, ticket / tail epilogue
    assert len(rows) >= 12, len(rows)
    for k in rows:
        assert kernels[k]["VGPRs"] <= 64 and kernels[k]["Occupancy [waves/SIMD]"] == 8 and kernels[k]["ScratchSize [bytes/lane]"] == 0, (k, kernels[k])
    # round 6: the three-trip ragged products at five wavefronts per SIMD without scratch (run-driven and list-driven windows, no window)
    for name in ("k_spmv_jagwI", "k_spmv_jaglI", "k_spmv_jagdI"):
        ks = [k for k in kernels if name in k]
        assert len(ks) >= 12, (name, len(ks))
        for k in ks:
            assert kernels[k]["VGPRs"] <= 96 and kernels[k]["ScratchSize [bytes/lane]"] == 0 and kernels[k]["Occupancy [waves/SIMD]"] >= 5, (k, kernels[k])
    pipe = [k for k in kernels if "k_spmm_pipeI" in k]
    assert len(pipe) == 4 and all(kernels[k]["ScratchSize [bytes/lane]"] <= 64 and kernels[k]["Occupancy [waves/SIMD]"] >= 2 for k in pipe), pipe
    # ... and its form for ragged rows: three workgroups of four wavefronts per CU, nothing in scratch (csrc/bicg_spmm_jag.hip)
    jpipe = [k for k in kernels if "k_spmm_jpipeI" in k]
    assert len(jpipe) == 6 and all(kernels[k]["ScratchSize [bytes/lane]"] == 0 and kernels[k]["Occupancy [waves/SIMD]"] >= 3 for k in jpipe), jpipe


def test_window_plan_does_not_depend_on_the_number_of_threads():
    """the set-up cuts its loops into one range per host thread (csrc/bicg_parallel.h): same runs, same slots for 1, 3 and 16"""
    A = synth.fem_like(n=117 * 117 * 3)
    L = H.lib()
    L.bicg_set_plan_threads.argtypes = [C.c_int]; L.bicg_set_plan_threads.restype = C.c_int
    before = -1                                        # back to the automatic count afterwards
    try:
        plans = []
        for nt in (1, 3, 16):
            assert L.bicg_set_plan_threads(nt) == nt
            plans.append(H.window_plan(A))
        for p in plans[1:]:
            assert np.array_equal(p[0], plans[0][0]) and np.array_equal(p[1], plans[0][1]) and p[2] == plans[0][2]
        # the plan of the persistent iteration (merged rows and padded slices filled by ranges of rows / workgroups)
        B = H.single_rank_blocks(synth.from_offsets(40013, (0, 1, -1, 117, -117, 118, -118, 3689, -3689, 3807, -3807), diag_base=14.0, seed=9))
        pp = []
        for nt in (1, 8):
            L.bicg_set_plan_threads(nt)
            pp.append(H.persist_plan(B, 1, 60))
        for key, v in pp[0].items():
            assert np.array_equal(v, pp[1][key]) if isinstance(v, np.ndarray) else v == pp[1][key], key
    finally:
        L.bicg_set_plan_threads(before)


def test_set_up_threads_are_the_host_threads_divided_by_the_ranks_of_the_node():
    """every rank's bicg_create used min(32, all hardware threads) whatever the number of ranks on the host.
    The automatic count is (threads this process may run on) / (ranks sharing the host), at most 32; the launcher's
    environment says how many ranks there are (torchrun's LOCAL_WORLD_SIZE here), the affinity mask how many threads."""
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from mpi_bicgstab_amd import hipsolver as H; L = H.lib(); "
            "L.bicg_set_plan_threads.restype = ctypes.c_int; print(L.bicg_set_plan_threads(0), len(os.sched_getaffinity(0)))" % ROOT)
    def ask(env_extra, cpus=None):
        env = {k: v for k, v in os.environ.items() if k not in ("BICG_PLAN_THREADS", "LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "SLURM_NTASKS_PER_NODE")}
        env.update(env_extra)
        cmd = [sys.executable, "-c", code]
        if cpus:
            cmd = ["taskset", "-c", cpus] + cmd
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=120, check=True).stdout.split()
        return int(out[0]), int(out[1])
    n1, hw = ask({})
    assert n1 == max(1, min(32, hw))
    n4, _ = ask({"LOCAL_WORLD_SIZE": "4"})
    assert n4 == max(1, min(32, hw // 4))
    n_env, _ = ask({"LOCAL_WORLD_SIZE": "4", "BICG_PLAN_THREADS": "3"})
    assert n_env == 3
    if hw >= 2 and shutil.which("taskset"):
        first_two = ",".join(str(c) for c in sorted(os.sched_getaffinity(0))[:2])
        n_pin, hw_pin = ask({}, cpus=first_two)
        assert hw_pin == 2 and n_pin == 2


def test_window_plan_covers_every_column_once():
    """bicg_window_plan (x windows of the ragged-rows SpMV): per 256-row group the runs
    are sorted, disjoint, cover every column the group touches, their slots are consecutive, gaps of <= 8 unused
    columns are bridged and nothing larger; bicg_window_slot inverts them; a group that touches too many
    scattered columns makes the whole plan fail."""
    A = synth.fem_like(n=117 * 117 * 3)
    plan = H.window_plan(A)
    assert plan is not None
    win_ptr, runs, used = plan
    ptr = A.ptr.astype(np.int64)
    ngroups = (A.rows + 255) // 256
    assert len(win_ptr) == ngroups + 1 and win_ptr[-1] == len(runs) and 0 < used <= 4096
    up = C.POINTER(C.c_uint)
    flat = np.ascontiguousarray(runs.reshape(-1))
    most = 0
    for g in (0, 1, 7, ngroups // 2, ngroups - 1):
        r0, r1 = g * 256, min(A.rows, g * 256 + 256)
        cols = np.unique(A.col[ptr[r0]:ptr[r1]]).astype(np.int64)
        rg = runs[win_ptr[g]:win_ptr[g + 1]].astype(np.int64)
        start, slot0, length = rg[:, 0], rg[:, 1] >> 16, rg[:, 1] & 0xFFFF
        assert np.all(np.diff(start) > 0) and np.all(start[1:] > start[:-1] + length[:-1] - 1 + 8)   # gaps > 8 between runs
        assert np.array_equal(slot0, np.concatenate(([0], np.cumsum(length)[:-1])))               # slots are consecutive
        covered = np.concatenate([np.arange(s, s + l) for s, l in zip(start, length)])
        assert np.all(np.isin(cols, covered))
        assert np.isin(start, cols).all() and np.isin(start + length - 1, cols).all()             # runs begin and end on a used column
        inside = np.isin(covered, cols)
        # a bridged gap is at most 8 unused columns long
        gaps = np.diff(np.flatnonzero(np.concatenate(([True], inside, [True])))) - 1
        assert gaps.max() <= 8
        for c in cols[:: max(1, len(cols) // 50)]:
            s = H.lib().bicg_window_slot(flat.ctypes.data_as(up), int(win_ptr[g]), int(win_ptr[g + 1]), int(c))
            i = np.searchsorted(start, c, side="right") - 1
            assert s == slot0[i] + (c - start[i]) and s < used
        most = max(most, int(length.sum()))
    assert most <= used
    assert used == 3 * (2 * 117 + 258)                    # the 27-point stencil: per z-plane the three y-neighbour clusters overlap
    B = synth.random_rows(20000, 40, seed=11)             # ~5000 scattered columns per group
    assert H.window_plan(B) is None
    assert H.window_plan(B, max_slots=65535) is not None


@pytest.mark.parametrize("world", [1, 3])
@pytest.mark.parametrize("kind", ["transport", "fem", "stencil"])
def test_persist_plan_decodes_back_to_the_matrix(kind, world):
    """bicg_persist_plan (the plan of the persistent iteration, host only): every workgroup owns spw
    consecutive 64-row slices; entry k of row r sits at pbase[slice] + 64 k + lane, diag entries first, then the offd entries
    in the [local rows | halo positions] numbering; its 16-bit slot, looked up in the workgroup's runs, is its column again;
    runs are ordered, their slots consecutive, none straddles the local / halo boundary; padding holds zeros; the largest
    window and matrix share are what the summary says."""
    if kind == "transport":
        A = synth.from_offsets(40013, (0, 1, -1, 117, -117, 118, -118, 3689, -3689, 3807, -3807), diag_base=14.0, seed=9)
    elif kind == "fem":
        A = synth.fem_like(n=117 * 117 * 2)
    else:
        A = synth.stencil7(23)
    for rank in range(world):
        diag, offd, counts, displs = synth.split_blocks(A, world, rank)
        blk = H.HostBlocks(diag, offd if world > 1 else None, A.rows, counts, displs)
        gmax = 60
        P = H.persist_plan(blk, world, gmax)
        assert P is not None
        n = diag.rows
        nslices = (n + 63) // 64
        assert P["rpt"] == 1 and P["spw"] == -(-nslices // gmax) and P["nwg"] == -(-nslices // P["spw"]) <= gmax
        halo = P["halo"]
        if world > 1:
            _, halo_cols, _, ren = H.halo_plan(blk, world)
        runs, wptr = P["runs"].astype(np.int64), P["wptr"]
        dptr, optr = diag.ptr.astype(np.int64), offd.ptr.astype(np.int64)
        most_slots = most_entries = 0
        for g in range(P["nwg"]):
            rg = runs[wptr[g]:wptr[g + 1]]
            first, slot0, length = rg[:, 0], rg[:, 1] >> 16, rg[:, 1] & 0xFFFF
            assert np.all(np.diff(first) > 0) and np.array_equal(slot0, np.concatenate(([0], np.cumsum(length)[:-1])))
            assert np.all((first + length <= n) | (first >= n)), "a run straddles the local / halo boundary"
            assert np.all(first + length <= n + halo)
            most_slots = max(most_slots, int(length.sum()))
            s0, s1 = g * P["spw"], min(nslices, (g + 1) * P["spw"])
            most_entries = max(most_entries, int(P["pbase"][s1] - P["pbase"][s0]))
            col_of_slot = np.concatenate([np.arange(f, f + l) for f, l in zip(first, length)]) if len(rg) else np.zeros(0, dtype=np.int64)
            for r in range(s0 * 64, min(n, s1 * 64)):
                want_cols = diag.col[dptr[r]:dptr[r + 1]].astype(np.int64)
                want_vals = diag.val[dptr[r]:dptr[r + 1]]
                if world > 1:
                    want_cols = np.concatenate((want_cols, ren[optr[r]:optr[r + 1]].astype(np.int64)))
                    want_vals = np.concatenate((want_vals, offd.val[optr[r]:optr[r + 1]]))
                assert P["rdiag"][r] == dptr[r + 1] - dptr[r] and P["rlen"][r] == len(want_cols)
                e = int(P["pbase"][r // 64]) + 64 * np.arange(len(want_cols)) + r % 64
                assert np.array_equal(col_of_slot[P["pslot"][e]], want_cols), (g, r)
                assert np.array_equal(P["pval"][e], want_vals)
        assert most_slots == P["win_slots"] and most_entries == P["max_entries"]
        # padding: everything that is not an entry of some row is zero
        total = sum(int(P["rlen"][r]) for r in range(n))
        assert np.count_nonzero(P["pval"]) <= total and P["entries"] >= total
    # more than 15 slices per workgroup: two rows per thread up to 30, eight (on 7 row wavefronts) up to 56 -- the plan is the
    # same structure with spw = row wavefronts x rows per thread; beyond that the block does not qualify
    big = H.single_rank_blocks(synth.from_offsets(64 * 16 * 3 + 1, (0, 1, -1), diag_base=4.0, seed=1))
    P2 = H.persist_plan(big, 1, 3)             # 49 slices over 3 workgroups: 17 each -> 9 row wavefronts x 2 rows
    assert P2 is not None and P2["rpt"] == 2 and P2["spw"] == 18 and P2["nwg"] == 3
    P8 = H.persist_plan(big, 1, 1)             # 49 slices in one workgroup -> 7 row wavefronts x 8 rows
    assert P8 is not None and P8["rpt"] == 8 and P8["spw"] == 56 and P8["nwg"] == 1
    huge = H.single_rank_blocks(synth.from_offsets(64 * 57 + 1, (0, 1, -1), diag_base=4.0, seed=1))
    assert H.persist_plan(huge, 1, 1) is None


def test_rccl_loaded_by_the_library_then_torch_exits_cleanly():
    """The RCCL transport dlopen()s librccl, which brings /opt/rocm's librocm_smi64 with it; PyTorch bundles its own
    copy. With the first made RTLD_GLOBAL a later `import torch` bound the second copy's statics to it and the process
    died at exit with "double free or corruption" -- exit status 134 from a test run in which everything had passed
    (tests/test_comm_path_one_gpu.py followed by any module that imports torch). bicg_comm_rccl_loadable resolves the
    entry points without a device call, so the load order can be exercised here."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mpi_bicgstab_amd import hipsolver as H\n"
            "assert H.lib().bicg_comm_rccl_loadable() == 1\n"
            "import torch\nprint('loaded both')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "loaded both" in out.stdout, (out.returncode, out.stderr[-400:])


def test_reference_shifted_driver_links_against_the_library():
    """oracle/_ref/shifted_dropin: the reference's main_shifted.c (driver of BASELINE.json configs[4]) with its own
    matrix.c / vector.c / mmio.c, the solver taken from libbicgstab_hip.so. Checked here without a GPU: it links, the only
    solver symbol it imports is the one src/main_shifted.c:126 calls, and the library defines nothing that the
    reference's own translation units define (no interposition of csr_*, my_*, MPI_csr_* by accident)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "shifted_dropin")
    lib = os.path.join(ROOT, "mpi-bicgstab_amd", "libbicgstab_hip.so")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    wanted = {ln.split()[-1] for ln in und.splitlines() if ln.split() and re.match(r"(shifted_|bicg|ca_bicg|pipe_bicg)", ln.split()[-1])}
    assert wanted == {"shifted_lopbicg_switching"}, wanted
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    lib_syms = {ln.split()[-1] for ln in exported.splitlines() if ln.split()}
    assert "shifted_lopbicg_switching" in lib_syms
    own = subprocess.run(["nm", "--defined-only", exe], capture_output=True, text=True, check=True).stdout
    exe_syms = {ln.split()[-1] for ln in own.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TD"}
    clash = {s for s in exe_syms & lib_syms if not s.startswith("_")}
    assert not clash, clash


def test_stencil7_closed_form_is_the_oracles_spmv():
    """synth.stencil7_matvec (no matrix: shifted grid views, stored-order sums) is bit-identical to the oracle's mult() on
    the assembled stencil -- the checker of the 512^3 legs, where no CPU oracle run fits (tests/test_full_size.py)"""
    import oracle_lib as O
    from mpi_bicgstab_amd import synth
    for m, w in ((12, (6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.0)), (17, synth.LAPLACE_WEIGHTS), (2, (3.0, 1.0, 2.0, 4.0, 5.0, 6.0, 7.0))):
        A = synth.stencil7(m, w)
        row, col, val = A.to_coo()
        for seed in (0, 1):
            x = np.random.default_rng(seed + m).standard_normal(A.rows)
            assert np.array_equal(synth.stencil7_matvec(m, w, x), O.spmv(A.rows, row, col, val, x))
        ones = synth.stencil7_matvec(m, synth.LAPLACE_WEIGHTS, np.ones(m ** 3)).reshape(m, m, m)
        assert np.all(ones[1:-1, 1:-1, 1:-1] == 0.0) and ones[0, 0, 0] == 3.0      # b = A 1: zero in the interior


def test_switch_variables_as_the_library_reads_them():
    """BICG_PLAN / BICG_PERSIST / BICG_TEST hold comma-separated `name` / `name=value` tokens (csrc/bicg_knobs.h): what
    hipsolver.switches() writes is what the library reads (bicg_switch_value, no device needed) -- whole names only, a bare token
    reads as "1", the last of the three stays readable while a caller holds the first."""
    from mpi_bicgstab_amd import hipsolver as H
    H.switches(stencil=0, lines=2, planes=64, layout="pad", persist=0, persist_chunk=17, force_comm=1, spin_ticks=0)
    assert os.environ["BICG_PLAN"] == "stencil=0,lines=2,planes=64,layout=pad" and os.environ["BICG_PERSIST"] == "0,chunk=17"
    assert H.switch_value("BICG_PLAN", "stencil") == "0" and H.switch_value("BICG_PLAN", "lines") == "2"
    assert H.switch_value("BICG_PLAN", "planes") == "64" and H.switch_value("BICG_PLAN", "layout") == "pad"
    assert H.switch_value("BICG_PLAN", "line") is None and H.switch_value("BICG_PLAN", "plane") is None      # prefixes are not names
    assert H.switch_value("BICG_PLAN", "ca-fuse") is None
    assert H.switch_value("BICG_PERSIST", "0") == "1" and H.switch_value("BICG_PERSIST", "chunk") == "17"
    assert H.switch_value("BICG_TEST", "force-comm") == "1" and H.switch_value("BICG_TEST", "spin-ticks") == "0"
    H.switches(persist=1, lines=None)
    assert H.switch_value("BICG_PERSIST", "0") is None and H.switch_value("BICG_PERSIST", "chunk") == "17"
    assert H.switch_value("BICG_PLAN", "lines") is None and H.switch_value("BICG_PLAN", "planes") == "64"
    os.environ["BICG_TEST"] = " plan-collide , p2p-fault-after=7,"            # written by hand: blanks, a bare token, a trailing comma
    assert H.switch_value("BICG_TEST", "plan-collide") == "1" and H.switch_value("BICG_TEST", "p2p-fault-after") == "7"
    # a token the library does not know is counted (bicg_create says so on rank 0) -- every keyword of hipsolver.switches is known
    buf = C.create_string_buffer(64)
    for var in H.SWITCH_VARS:
        assert H.lib().bicg_switch_unknown(var.encode(), buf, 64) == 0, (var, os.environ.get(var), buf.value)
    everything = {k: 1 for k in H.SWITCHES}
    everything["layout"] = "jag"
    H.switches(**everything)
    for var in H.SWITCH_VARS:
        assert H.lib().bicg_switch_unknown(var.encode(), buf, 64) == 0, (var, os.environ.get(var), buf.value)
    H.switches(**{k: None for k in H.SWITCHES})
    os.environ["BICG_PLAN"] = "stencil=0,stencl=0,no-window"
    assert H.lib().bicg_switch_unknown(b"BICG_PLAN", buf, 64) == 2 and buf.value == b"stencl=0"
    # ... and so is a known name with a value that cannot be read: values are integers or on / off / true / false / yes / no
    # (layout: jag / pad); a switch spelled out reads as 1 / 0
    os.environ["BICG_PLAN"] = "stencil=on,jagw=false,ca-fuse=yes,window=-1,layout=pad"
    assert H.lib().bicg_switch_unknown(b"BICG_PLAN", buf, 64) == 0, buf.value
    assert H.switch_value("BICG_PLAN", "stencil") == "1" and H.switch_value("BICG_PLAN", "jagw") == "0" and H.switch_value("BICG_PLAN", "ca-fuse") == "1"
    for text, n, first in (("lines=2;planes=8", 1, b"lines=2;planes=8"), ("stencil = 0", 2, b"="), ("stencil=O", 1, b"stencil=O"),
                           ("layout=jagged,planes=", 2, b"layout=jagged"), ("lines=2.5", 1, b"lines=2.5")):
        os.environ["BICG_PLAN"] = text
        assert H.lib().bicg_switch_unknown(b"BICG_PLAN", buf, 64) == n and buf.value == first, (text, buf.value)
    # the persistent forms are the default: BICG_PERSIST knows "0" / "off" and its two valued tokens, nothing that would switch them "on"
    os.environ["BICG_PERSIST"] = "1"
    assert H.lib().bicg_switch_unknown(b"BICG_PERSIST", buf, 64) == 1
    os.environ["BICG_PERSIST"] = "off,chunk=64"
    assert H.lib().bicg_switch_unknown(b"BICG_PERSIST", buf, 64) == 0
    manual = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "`0` (or `off`)" in manual and "BICG_PERSIST=1" not in manual and "BICG_PERSIST=on" not in manual
    del os.environ["BICG_PERSIST"]
    os.environ["BICG_PLAN"] = "lines=2"
    H.switches(stencil=None, lines=None, planes=None, layout=None, persist_chunk=None)
    assert "BICG_PLAN" not in os.environ and "BICG_PERSIST" not in os.environ
    assert H.switch_value("BICG_PLAN", "stencil") is None


def test_the_default_build_reads_at_most_thirty_variables_and_the_manual_lists_them():
    """every variable the default build reads (getenv directly, or one of the three token lists) is in INTEGRATION.md section 6;
    the measurement knobs of the development rounds (knob_x) are not read without -DBICG_EXPERIMENTS (csrc/bicg_knobs.h)"""
    import glob
    import re
    pkg = os.path.join(ROOT, "mpi-bicgstab_amd")
    names = set()
    for path in glob.glob(os.path.join(pkg, "csrc", "*")) + glob.glob(os.path.join(pkg, "host", "*.c")) + glob.glob(os.path.join(pkg, "host", "*.h")):
        text = open(path, errors="replace").read()
        names.update(re.findall(r'[^_a-z]getenv\("(BICG_[A-Z0-9_]+)"\)', text))
        names.update(re.findall(r'knob_tok\("(BICG_[A-Z0-9_]+)"', text))
    assert {"BICG_PLAN", "BICG_PERSIST", "BICG_TEST"} <= names
    assert len(names) <= 30, sorted(names)
    manual = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in sorted(names) if f"`{n}`" not in manual and f"`{n}=" not in manual]
    assert not missing, missing
