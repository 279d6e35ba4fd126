"""The C host's Matrix-Market block loader (mpi-bicgstab_amd/host/bicg_mtx.c), CPU only.

Serial mode (every rank reads the file) and MPI mode (every rank tokenises 1/P of the bytes, then
MPI_Alltoallv) must both give what the reference's MPI_csr_load_matrix_block gives: the equal-rows
partition, diag block with local columns, offd block with global columns, and inside a row the
FILE order (reference src/matrix.c:135-183, 295-308, 336-392)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from mpi_bicgstab_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "mpi-bicgstab_amd", "host", "bicg_mtx_dump")
MPIEXEC = "/opt/conda/bin/mpiexec"

pytestmark = pytest.mark.skipif(not (os.path.exists(DUMP) and os.path.exists(MPIEXEC)), reason="host tools / MPI not built")


def _read(prefix, rank):
    raw = open(f"{prefix}.rank{rank}.bin", "rb").read()
    rows, ncols, nd, no = struct.unpack("4I", raw[:16])
    off = 16

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=off)
        off += a.nbytes
        return a
    d = (take(np.uint32, rows + 1), take(np.uint32, nd), take(np.float64, nd))
    o = (take(np.uint32, rows + 1), take(np.uint32, no), take(np.float64, no))
    return rows, ncols, d, o


def _read_partition(prefix, rank, world):
    raw = open(f"{prefix}.rank{rank}.bin", "rb").read()
    tail = np.frombuffer(raw[-8 * world:], dtype=np.int32)
    return tail[:world].copy(), tail[world:].copy()      # displs, counts


def _expected(n, row, col, val, world, rank):
    """stable sort of the file-order triplets by row, then the reference's diag/offd split"""
    order = np.argsort(row, kind="stable")
    r, c, v = row[order], col[order], val[order]
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ptr, r.astype(np.int64) + 1, 1)
    ptr = np.cumsum(ptr)
    A = synth.CSR(n, n, ptr.astype(np.uint32), c.astype(np.uint32), v)
    return synth.split_blocks(A, world, rank)


@pytest.mark.parametrize("order", ["colmajor", "shuffled"])
@pytest.mark.parametrize("world,mode", [(1, "serial"), (3, "serial"), (2, "mpi"), (3, "mpi"), (5, "mpi")])
def test_loader_blocks(tmp_path, world, mode, order):
    A = synth.from_offsets(997, (0, 1, -1, 30, -30, 400, -400), diag_base=7.0, seed=3)
    row, col, val = synth.colmajor_coo(A)
    if order == "shuffled":
        perm = np.random.default_rng(5).permutation(len(val))
        row, col, val = row[perm], col[perm], val[perm]
    mtx = str(tmp_path / "m.mtx")
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% a comment line\n")
        f.write(f"{A.rows} {A.cols} {A.nnz}\n")
        for i, j, v in zip(row.tolist(), col.tolist(), val.tolist()):
            f.write(f"{i + 1} {j + 1} {v!r}\n")
    prefix = str(tmp_path / "out")
    # the entry lines (serial mode: all of them; MPI mode: this rank's byte range) tokenised by one thread, and by several
    # (sub-ranges cut at line boundaries, lists taken in range order: the file order inside every row must survive)
    for threads in ("1", "5"):
        subprocess.run([MPIEXEC, "-n", str(world), DUMP, mtx, prefix, mode], check=True, timeout=120,
                       env=dict(os.environ, BICG_MTX_THREADS=threads))
        for rank in range(world):
            rows, ncols, d, o = _read(prefix, rank)
            ed, eo, counts, _ = _expected(A.rows, row, col, val, world, rank)
            assert rows == counts[rank] and ncols == A.cols
            assert np.array_equal(d[0], ed.ptr) and np.array_equal(d[1], ed.col) and np.array_equal(d[2], ed.val), threads
            assert np.array_equal(o[0], eo.ptr) and np.array_equal(o[1], eo.col) and np.array_equal(o[2], eo.val), threads


def test_loader_symmetric_and_pattern(tmp_path):
    """beyond the reference: symmetric storage is mirrored, pattern entries become 1.0
    (the reference's block loader does neither, SURVEY.md section 4 defect 2)"""
    mtx = str(tmp_path / "s.mtx")
    open(mtx, "w").write("%%MatrixMarket matrix coordinate pattern symmetric\n4 4 5\n1 1\n2 1\n3 2\n4 4\n4 1\n")
    prefix = str(tmp_path / "o")
    subprocess.run([MPIEXEC, "-n", "2", DUMP, mtx, prefix, "mpi"], check=True, timeout=60)
    dense = np.zeros((4, 4))
    for rank, lo in ((0, 0), (1, 2)):
        rows, ncols, d, o = _read(prefix, rank)
        for i in range(rows):
            for k in range(d[0][i], d[0][i + 1]):
                dense[lo + i, lo + d[1][k]] += d[2][k]
            for k in range(o[0][i], o[0][i + 1]):
                dense[lo + i, o[1][k]] += o[2][k]
    want = np.zeros((4, 4))
    for i, j in ((0, 0), (1, 0), (2, 1), (3, 3), (3, 0)):
        want[i, j] = 1.0
        want[j, i] = 1.0
    assert np.array_equal(dense, want)


def test_symmetric_nz_counts_mirrored_entries_and_skew_is_refused(tmp_path):
    """INFO_Matrix.nz is the number of entries of the matrix that is solved (8 here: 5 stored, 3 mirrored), not the
    banner's triangle count, in the serial and in the MPI loader; skew-symmetric / hermitian files are refused
    instead of being read as general"""
    import ctypes as C
    from mpi_bicgstab_amd import hipsolver as H
    mtx = str(tmp_path / "s.mtx")
    open(mtx, "w").write("%%MatrixMarket matrix coordinate real symmetric\n4 4 5\n1 1 2.0\n2 1 -1.0\n3 2 -1.0\n4 4 2.0\n4 1 0.5\n")
    blk = H.load_mtx_blocks(mtx, 0, 1)
    assert blk.nnz_global == 8 and int(blk.diag.nz) == 8
    out = subprocess.run([MPIEXEC, "-n", "2", DUMP, mtx, str(tmp_path / "o"), "mpi"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "mirrored to 8 entries" in out.stderr
    for kind in ("skew-symmetric", "hermitian"):
        bad = str(tmp_path / f"{kind}.mtx")
        open(bad, "w").write(f"%%MatrixMarket matrix coordinate real {kind}\n2 2 1\n2 1 1.0\n")
        d, o, info = H.CSRMatrix(), H.CSRMatrix(), H.InfoMatrix()
        L = H.lib()
        L.bicg_mtx_load_block.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(H.CSRMatrix), C.POINTER(H.CSRMatrix), C.POINTER(H.InfoMatrix)]
        assert L.bicg_mtx_load_block(bad.encode(), 0, 1, C.byref(d), C.byref(o), C.byref(info)) != 0


def _write_mtx(path, A, row, col, val):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{A.rows} {A.cols} {len(val)}\n")
        for i, j, v in zip(row.tolist(), col.tolist(), val.tolist()):
            f.write(f"{i + 1} {j + 1} {v!r}\n")


def _blocks_for(A, displs, counts, rank):
    """diag/offd split of CSR A (file order inside rows already) for an arbitrary contiguous partition"""
    lo, hi = int(displs[rank]), int(displs[rank] + counts[rank])
    ptr = A.ptr.astype(np.int64)
    dptr, optr, dcol, ocol, dval, oval = [0], [0], [], [], [], []
    for r in range(lo, hi):
        for k in range(ptr[r], ptr[r + 1]):
            c = int(A.col[k])
            if lo <= c < hi:
                dcol.append(c - lo); dval.append(A.val[k])
            else:
                ocol.append(c); oval.append(A.val[k])
        dptr.append(len(dcol)); optr.append(len(ocol))
    return (np.array(dptr), np.array(dcol, dtype=np.int64), np.array(dval)), (np.array(optr), np.array(ocol, dtype=np.int64), np.array(oval))


@pytest.mark.parametrize("world,mode", [(1, "serial"), (3, "serial"), (2, "mpi"), (4, "mpi")])
def test_loader_nnz_balanced_partition_and_cache(tmp_path, world, mode):
    """non-zero balanced row blocks (the reference's abandoned DYNAMIC_ROWS idea,
    archive/matrix.c:407-446) and the binary block cache round trip. A ragged matrix (row lengths
    0..60 plus one row of 400) makes equal-rows blocks badly unbalanced."""
    A = synth.random_rows(1500, 60, seed=12, long_rows={700: 400})
    row, col, val = A.to_coo()
    mtx = str(tmp_path / "r.mtx")
    _write_mtx(mtx, A, row, col, val)
    prefix = str(tmp_path / "out")
    cache = tmp_path / "cache"
    cache.mkdir()
    subprocess.run([MPIEXEC, "-n", str(world), DUMP, mtx, prefix, mode, "nnz", str(cache)], check=True, timeout=120,
                   env=dict(os.environ, BICG_MTX_THREADS="3"))
    lens = np.diff(A.ptr.astype(np.int64))
    displs, counts = _read_partition(prefix, 0, world)
    assert displs[0] == 0 and np.array_equal(displs[1:], np.cumsum(counts)[:-1]) and counts.sum() == A.rows
    per_rank = np.array([lens[displs[p]:displs[p] + counts[p]].sum() for p in range(world)])
    # every block within one (longest) row of the ideal share
    assert np.abs(per_rank - lens.sum() / world).max() <= lens.max(), (per_rank, lens.sum() / world)
    for rank in range(world):
        d2, c2 = _read_partition(prefix, rank, world)
        assert np.array_equal(d2, displs) and np.array_equal(c2, counts)        # same cuts on every rank
        rows, ncols, d, o = _read(prefix, rank)                                   # (after the cache round trip)
        ed, eo = _blocks_for(A, displs, counts, rank)
        assert rows == counts[rank]
        assert np.array_equal(d[0], ed[0]) and np.array_equal(d[1], ed[1]) and np.array_equal(d[2], ed[2])
        assert np.array_equal(o[0], eo[0]) and np.array_equal(o[1], eo[1]) and np.array_equal(o[2], eo[2])
    assert len(list(cache.iterdir())) == world


def test_cache_rejects_stale_and_corrupt(tmp_path):
    import ctypes as C
    from mpi_bicgstab_amd import hipsolver as H
    L = H.lib()
    A = synth.from_offsets(300, (0, 1, -1, 17, -17), diag_base=5.0, seed=1)
    row, col, val = synth.colmajor_coo(A)
    mtx = str(tmp_path / "m.mtx")
    _write_mtx(mtx, A, row, col, val)
    d, o, info = H.CSRMatrix(), H.CSRMatrix(), H.InfoMatrix()
    L.bicg_mtx_load_block_part.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(H.CSRMatrix), C.POINTER(H.CSRMatrix), C.POINTER(H.InfoMatrix)]
    args6 = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(H.CSRMatrix), C.POINTER(H.CSRMatrix), C.POINTER(H.InfoMatrix)]
    L.bicg_mtx_cache_save.argtypes = args6
    L.bicg_mtx_cache_load.argtypes = args6
    assert L.bicg_mtx_load_block_part(mtx.encode(), 1, 2, 0, C.byref(d), C.byref(o), C.byref(info)) == 0
    cpath = str(tmp_path / "c.bicgblk").encode()
    assert L.bicg_mtx_cache_save(cpath, mtx.encode(), 1, 2, 0, C.byref(d), C.byref(o), C.byref(info)) == 0
    d2, o2, i2 = H.CSRMatrix(), H.CSRMatrix(), H.InfoMatrix()
    assert L.bicg_mtx_cache_load(cpath, mtx.encode(), 1, 2, 0, C.byref(d2), C.byref(o2), C.byref(i2)) == 0
    assert d2.rows == d.rows and d2.ptr[d2.rows] == d.ptr[d.rows]
    assert L.bicg_mtx_cache_load(cpath, mtx.encode(), 0, 2, 0, C.byref(d2), C.byref(o2), C.byref(i2)) != 0     # other rank
    # a flipped byte in the payload fails the checksum
    raw = bytearray(open(cpath.decode(), "rb").read())
    raw[len(raw) // 2] ^= 0x40
    open(cpath.decode(), "wb").write(bytes(raw))
    assert L.bicg_mtx_cache_load(cpath, mtx.encode(), 1, 2, 0, C.byref(d2), C.byref(o2), C.byref(i2)) != 0
    # source file changed after the cache was written
    assert L.bicg_mtx_cache_save(cpath, mtx.encode(), 1, 2, 0, C.byref(d), C.byref(o), C.byref(info)) == 0
    with open(mtx, "a") as f:
        f.write("% touched\n")
    assert L.bicg_mtx_cache_load(cpath, mtx.encode(), 1, 2, 0, C.byref(d2), C.byref(o2), C.byref(i2)) != 0


def test_block_builder_hook(tmp_path):
    """bicg_mtx_set_block_builder: the loader hands a rank's triplets (file order, global indices) to the
    installed builder (on the GPU box: bicg_coo_to_blocks_device) and uses the blocks it returns. Here the
    builder is a Python callback that does the stable row sort with numpy and allocates with malloc."""
    import ctypes as C
    from mpi_bicgstab_amd import hipsolver as H
    L = H.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    up, dp = C.POINTER(C.c_uint), C.POINTER(C.c_double)
    BUILDER = C.CFUNCTYPE(C.c_int, up, up, dp, C.c_ulong, C.c_uint, C.c_uint, C.c_uint, C.POINTER(H.CSRMatrix), C.POINTER(H.CSRMatrix))
    seen = []

    def fill(dst, rows, cols, r, c, v):
        ptr = np.zeros(rows + 1, dtype=np.uint32)
        np.add.at(ptr, r.astype(np.int64) + 1, 1)
        ptr = np.cumsum(ptr).astype(np.uint32)
        order = np.argsort(r, kind="stable")
        c, v = np.ascontiguousarray(c[order], dtype=np.uint32), np.ascontiguousarray(v[order], dtype=np.float64)
        for name, arr in (("ptr", ptr), ("col", c), ("val", v)):
            mem = libc.malloc(max(arr.nbytes, 8))
            C.memmove(mem, arr.ctypes.data, arr.nbytes)
            setattr(dst.contents, name, C.cast(mem, up if name != "val" else dp))
        dst.contents.rows, dst.contents.cols, dst.contents.nz = rows, cols, len(v)

    def builder(row, col, val, nnz, lo, hi, ncols, diag, offd):
        r = np.ctypeslib.as_array(row, shape=(nnz,)).copy(); c = np.ctypeslib.as_array(col, shape=(nnz,)).copy()
        v = np.ctypeslib.as_array(val, shape=(nnz,)).copy()
        seen.append((int(nnz), int(lo), int(hi)))
        local = (c >= lo) & (c < hi)
        fill(diag, hi - lo, hi - lo, r[local] - lo, c[local] - lo, v[local])
        fill(offd, hi - lo, ncols, r[~local] - lo, c[~local], v[~local])
        return 0

    cb = BUILDER(builder)
    A = synth.from_offsets(500, (0, 1, -1, 23, -23, 200, -200), diag_base=6.0, seed=2)
    row, col, val = synth.colmajor_coo(A)
    mtx = str(tmp_path / "h.mtx")
    _write_mtx(mtx, A, row, col, val)
    L.bicg_mtx_load_block.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(H.CSRMatrix), C.POINTER(H.CSRMatrix), C.POINTER(H.InfoMatrix)]
    L.bicg_mtx_set_block_builder.argtypes = [C.c_void_p]
    plain = (H.CSRMatrix(), H.CSRMatrix(), H.InfoMatrix())
    assert L.bicg_mtx_load_block(mtx.encode(), 1, 3, *map(C.byref, plain)) == 0
    L.bicg_mtx_set_block_builder(C.cast(cb, C.c_void_p))
    try:
        hooked = (H.CSRMatrix(), H.CSRMatrix(), H.InfoMatrix())
        assert L.bicg_mtx_load_block(mtx.encode(), 1, 3, *map(C.byref, hooked)) == 0
    finally:
        L.bicg_mtx_set_block_builder(None)
    assert len(seen) == 1 and seen[0][1:] == (167, 334)
    for a, b in zip(plain[:2], hooked[:2]):
        n = a.rows
        assert b.rows == n and b.cols == a.cols
        pa, pb = np.ctypeslib.as_array(a.ptr, shape=(n + 1,)), np.ctypeslib.as_array(b.ptr, shape=(n + 1,))
        assert np.array_equal(pa, pb)
        nz = int(pa[-1])
        if nz:
            assert np.array_equal(np.ctypeslib.as_array(a.col, shape=(nz,)), np.ctypeslib.as_array(b.col, shape=(nz,)))
            assert np.array_equal(np.ctypeslib.as_array(a.val, shape=(nz,)), np.ctypeslib.as_array(b.val, shape=(nz,)))


def test_entry_count_and_index_range_do_not_depend_on_threads(tmp_path):
    """The reference reads exactly the banner's nz entry lines (src/matrix.c:315-331: fewer is "ERROR: reading matrix data",
    lines after the nz-th are never looked at) and would index outside its arrays for an entry outside the matrix. Here: the
    serial-mode loader keeps the FIRST nz lines whatever the number of tokeniser threads, a short file fails with the
    reference's message, an index outside m x n (0 included) is an error everywhere; the MPI byte-range mode finds the nz-th
    line with a prefix sum over the ranks and drops what lies behind it."""
    A = synth.from_offsets(300, (0, 1, -1, 17, -17), diag_base=5.0, seed=2)
    row, col, val = synth.colmajor_coo(A)
    lines = [f"{i + 1} {j + 1} {v!r}\n" for i, j, v in zip(row.tolist(), col.tolist(), val.tolist())]
    extra = ["1 1 99.5\n", "300 300 -3.0\n", "2 1 7.0\n"]

    def write(name, body, nz=A.nnz):
        path = str(tmp_path / name)
        with open(path, "w") as f:
            f.write("%%MatrixMarket matrix coordinate real general\n")
            f.write(f"{A.rows} {A.cols} {nz}\n")
            f.writelines(body)
        return path

    def run(path, world, mode, threads):
        return subprocess.run([MPIEXEC, "-n", str(world), DUMP, path, str(tmp_path / "o"), mode], capture_output=True, text=True, timeout=120,
                              env=dict(os.environ, BICG_MTX_THREADS=threads))
    more = write("more.mtx", lines + extra)
    for threads in ("1", "4", "16"):
        out = run(more, 2, "serial", threads)
        assert out.returncode == 0, out.stderr
        for rank in range(2):
            rows, ncols, d, o = _read(str(tmp_path / "o"), rank)
            ed, eo, counts, _ = _expected(A.rows, row, col, val, 2, rank)
            assert np.array_equal(d[0], ed.ptr) and np.array_equal(d[2], ed.val) and np.array_equal(o[2], eo.val), threads
    # the byte-range (MPI) mode: a prefix sum of the ranks' line counts finds the nz-th line; what lies behind it is dropped
    for world, threads in ((2, "2"), (4, "1"), (3, "5")):
        out = run(more, world, "mpi", threads)
        assert out.returncode == 0, out.stderr
        for rank in range(world):
            rows, ncols, d, o = _read(str(tmp_path / "o"), rank)
            ed, eo, counts, _ = _expected(A.rows, row, col, val, world, rank)
            assert np.array_equal(d[0], ed.ptr) and np.array_equal(d[2], ed.val) and np.array_equal(o[2], eo.val), (world, threads)
    # garbage and out-of-range lines BEHIND the nz-th entry line are never read by the reference: no error, no message,
    # whatever the number of tokeniser threads
    junk = write("junk.mtx", lines + ["999 999 1.0\n", "this is not an entry\n", "0 0 0\n"] * 40)
    for threads in ("1", "4", "16"):
        out = run(junk, 2, "serial", threads)
        assert out.returncode == 0 and "ERROR" not in out.stderr, (threads, out.stderr)
        for rank in range(2):
            rows, ncols, d, o = _read(str(tmp_path / "o"), rank)
            ed, eo, counts, _ = _expected(A.rows, row, col, val, 2, rank)
            assert np.array_equal(d[0], ed.ptr) and np.array_equal(d[2], ed.val) and np.array_equal(o[2], eo.val), threads
    # ... and in the byte-range (MPI) mode: the first range that does not parse knows how many entries lie in front
    # of it; nz or more, and it is never looked at
    for world, threads in ((2, "2"), (3, "1"), (4, "3"), (8, "1")):
        out = run(junk, world, "mpi", threads)
        assert out.returncode == 0 and "ERROR" not in out.stderr, (world, threads, out.stderr)
        for rank in range(world):
            rows, ncols, d, o = _read(str(tmp_path / "o"), rank)
            ed, eo, counts, _ = _expected(A.rows, row, col, val, world, rank)
            assert np.array_equal(d[0], ed.ptr) and np.array_equal(d[2], ed.val) and np.array_equal(o[2], eo.val), (world, threads)
    # a malformed line among the FIRST nz entries is an error in every mode, on every rank (no rank left behind in a collective)
    broken = write("broken.mtx", lines[:200] + ["this is not an entry\n"] + lines[200:])
    for mode, world in (("serial", 2), ("mpi", 2), ("mpi", 5)):
        out = run(broken, world, mode, "2")
        assert out.returncode != 0 and "ERROR: reading matrix data" in out.stderr, (mode, world, out.stderr)
    short = write("short.mtx", lines[:-5])
    for mode, threads in (("serial", "1"), ("serial", "4"), ("mpi", "2")):
        out = run(short, 2, mode, threads)
        assert out.returncode != 0 and "ERROR: reading matrix data" in out.stderr, (mode, threads, out.stderr)
    for bad in ("0 5 1.0\n", "301 1 1.0\n", "4 301 1.0\n"):
        path = write("bad.mtx", lines[:100] + [bad] + lines[101:])
        for mode, threads in (("serial", "1"), ("serial", "4"), ("mpi", "2")):
            out = run(path, 2, mode, threads)
            assert out.returncode != 0 and "outside the 300 x 300 matrix" in out.stderr, (bad, mode, threads, out.stderr)


def test_mpi_mode_with_more_ranks_than_entry_lines(tmp_path):
    """Eight ranks, six entry lines: some byte ranges hold no line start. Every collective of the MPI-mode loader is called
    by every rank (round 4's loader called MPI_Comm_split_type only on ranks WITH lines and hung here)."""
    n = 6
    row = np.arange(n, dtype=np.int64)
    col = (np.arange(n, dtype=np.int64) * 5) % n
    val = np.linspace(1.0, 2.0, n)
    mtx = str(tmp_path / "six.mtx")
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n} {n}\n")
        for i, j, v in zip(row.tolist(), col.tolist(), val.tolist()):
            f.write(f"{i + 1} {j + 1} {v!r}\n")
    prefix = str(tmp_path / "out")
    for world in (8, 7):
        subprocess.run([MPIEXEC, "-n", str(world), DUMP, mtx, prefix, "mpi"], check=True, timeout=60)
        for rank in range(world):
            rows, ncols, d, o = _read(prefix, rank)
            ed, eo, counts, _ = _expected(n, row, col, val, world, rank)
            assert rows == counts[rank]
            assert np.array_equal(d[0], ed.ptr) and np.array_equal(d[1], ed.col) and np.array_equal(d[2], ed.val)
            assert np.array_equal(o[0], eo.ptr) and np.array_equal(o[1], eo.col) and np.array_equal(o[2], eo.val)


@pytest.mark.parametrize("world", [1, 2])
def test_loader_more_threads_than_rows_and_duplicates(tmp_path, world):
    """The serial-mode loader reads, tokenises and assembles with BICG_MTX_THREADS workers: byte ranges for the first two,
    ROW ranges for the assembly (a thread keeps what falls into its rows, walking the per-thread triplet lists in file
    order). Edge of that scheme: more workers than rows, rows without entries, repeated (row, col) pairs -- which the
    reference keeps as separate entries in file order (src/matrix.c:357-393)."""
    n = 5
    row = np.array([4, 0, 4, 2, 0, 4, 2, 0], dtype=np.int64)
    col = np.array([0, 3, 4, 2, 3, 0, 0, 0], dtype=np.int64)          # (0,3) and (4,0) twice; rows 1 and 3 empty
    val = np.array([1.5, -2.0, 3.25, 4.0, 5.0, 6.0, 7.0, 8.0])
    mtx = str(tmp_path / "tiny.mtx")
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n} {len(val)}\n")
        for i, j, v in zip(row.tolist(), col.tolist(), val.tolist()):
            f.write(f"{i + 1} {j + 1} {v!r}\n")
    prefix = str(tmp_path / "out")
    for threads in ("1", "3", "16"):
        subprocess.run([MPIEXEC, "-n", str(world), DUMP, mtx, prefix, "serial"], check=True, timeout=120,
                       env=dict(os.environ, BICG_MTX_THREADS=threads))
        for rank in range(world):
            rows, ncols, d, o = _read(prefix, rank)
            ed, eo, counts, _ = _expected(n, row, col, val, world, rank)
            assert rows == counts[rank] and ncols == n
            assert np.array_equal(d[0], ed.ptr) and np.array_equal(d[1], ed.col) and np.array_equal(d[2], ed.val), threads
            assert np.array_equal(o[0], eo.ptr) and np.array_equal(o[1], eo.col) and np.array_equal(o[2], eo.val), threads


def test_value_conversion_is_correctly_rounded():
    """The tokeniser converts values with its own routines (host/bicg_mtx.c: Clinger's exact case, Eisel-Lemire with a
    truncated 128-bit power-of-ten table, strtod for whatever those decline) where the reference uses fscanf("%lg")
    (src/matrix.c:333, 366). Every spelling must give the correctly rounded double -- bit for bit what Python's float()
    gives -- and stop at the same character as strtod: random doubles in several print formats, random digit strings over
    the whole exponent range, the neighbourhoods of half-way points between adjacent doubles (where a conversion that is
    merely accurate goes wrong), and the spellings only libc understands."""
    import ctypes as C
    import random
    import struct
    from decimal import Decimal, getcontext
    from mpi_bicgstab_amd import hipsolver as H
    L = H.lib()
    L.bicg_mtx_parse_double.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    v, n = C.c_double(), C.c_int()
    handled = {"own": 0, "libc": 0}

    def check(s, consumed=None, want=None):
        rc = L.bicg_mtx_parse_double(s.encode() + b"\n9 9 9.5\n", C.byref(v), C.byref(n))
        handled["own" if rc == 0 else "libc"] += 1
        used = len(s) if consumed is None else consumed
        assert n.value == used, (s, n.value, used)
        w = float(s[:used]) if want is None else want
        assert struct.pack("<d", v.value) == struct.pack("<d", w), (s, v.value, w, rc)
        return rc

    rng = random.Random(1234)
    # 1. random finite doubles, printed the ways matrix files are written
    for _ in range(60000):
        bits = rng.getrandbits(64)
        if (bits >> 52) & 0x7FF == 0x7FF:
            continue
        d = struct.unpack("<d", struct.pack("<Q", bits))[0]
        for fmt in ("%r", "%.17g", "%.16e", "%.15g", "%.9g"):
            check(fmt % d)
    # ... and doubles of ordinary magnitude (what a matrix holds)
    own_before = dict(handled)
    for _ in range(60000):
        d = rng.uniform(-1, 1) * 10.0 ** rng.randint(-12, 12)
        for fmt in ("%r", "%.17g", "%.16e", "%.12f", "%.6g"):
            check(fmt % d)
    own = handled["own"] - own_before["own"]
    assert own >= 0.999 * 300000, handled          # the fast routines, not libc, carry a real file
    # 2. random digit strings: 1..19 digits, a point anywhere, exponents over the whole range
    for _ in range(120000):
        nd = rng.randint(1, 19)
        digits = "".join(rng.choice("0123456789") for _ in range(nd))
        pt = rng.randint(0, nd)
        body = digits[:pt] + ("." if rng.random() < 0.7 else "") + digits[pt:]
        if body in (".", ""):
            continue
        if "." not in body and pt < nd:
            body = digits
        s = rng.choice(("", "-", "+")) + body
        if rng.random() < 0.8:
            s += rng.choice("eE") + rng.choice(("", "-", "+")) + str(rng.randint(0, 330))
        check(s)
    # 3. half-way neighbourhoods: m = (d + next(d)) / 2 exactly, written with 17..19 digits, and one unit either side
    getcontext().prec = 800
    for _ in range(20000):
        bits = rng.getrandbits(63) & ~(0x7FF << 52) | (rng.randint(1, 2045) << 52)
        d = struct.unpack("<d", struct.pack("<Q", bits))[0]
        up = struct.unpack("<d", struct.pack("<Q", bits + 1))[0]
        mid = (Decimal(d) + Decimal(up)) / 2
        for nd in (17, 18, 19):
            t = mid.as_tuple()
            lead = int("".join(map(str, t.digits[:nd])))
            e10 = len(t.digits) + t.exponent - nd
            for delta in (-1, 0, 1):
                check(f"{lead + delta}e{e10}")
    # 3b. more than 19 digits: 20..40 random digits, and half-way points written out to 25 / 40 digits (and one unit either
    #     side of them): decided by the routine only when both ends of the truncation interval round alike, else libc
    for _ in range(30000):
        nd = rng.randint(20, 40)
        digits = str(rng.randint(1, 9)) + "".join(rng.choice("0123456789") for _ in range(nd - 1))
        pt = rng.randint(0, nd)
        s = rng.choice(("", "-")) + digits[:pt] + "." + digits[pt:] + "e" + str(rng.randint(-320, 300))
        check(s)
    for _ in range(5000):
        bits = rng.getrandbits(63) & ~(0x7FF << 52) | (rng.randint(1, 2045) << 52)
        d = struct.unpack("<d", struct.pack("<Q", bits))[0]
        up = struct.unpack("<d", struct.pack("<Q", bits + 1))[0]
        t = ((Decimal(d) + Decimal(up)) / 2).as_tuple()
        for nd in (25, 40):
            dg = list(t.digits[:nd]) + [0] * max(0, nd - len(t.digits))
            lead = int("".join(map(str, dg)))
            e10 = len(t.digits) + t.exponent - nd
            for delta in (-1, 0, 1):
                check(f"{lead + delta}e{e10}")
    # 4. particular spellings
    for s in ("0", "-0", "0.0", "-0.0e5", "00012.5000", ".5", "5.", "+1.5", "1E+05", "1e22", "1e23", "1e-22", "8.5e-23",
              "9007199254740993", "9007199254740992", "4503599627370497.5", "123456789012345678e-5", "1.7976931348623157e308",
              "2.2250738585072014e-308", "2.2250738585072011e-308", "4.9e-324", "1e-330", "1e400", "-1e400",
              "12345678901234567890", "0.000000000000000000000000000000000012345678901234567890123"):
        check(s)
    check("1e", consumed=1)
    check("1e+", consumed=1)
    check("2.5e-x", consumed=3)
    check("7.25 ", consumed=4)
    check("  \t3.5", want=3.5)
    check("0x10", want=16.0)                 # strtod reads hexadecimal floats; so does the loader, through strtod
    check("inf", want=float("inf"))
    rc = L.bicg_mtx_parse_double(b"nan", C.byref(v), C.byref(n))
    assert rc == 1 and v.value != v.value and n.value == 3
    assert handled["own"] > 10 * handled["libc"], handled


REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")


@pytest.mark.skipif(not os.path.exists(REF_DUMP), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("world", [1, 2, 3])
def test_loader_against_the_reference_loader_itself(tmp_path, world):
    """The pin of SURVEY.md section 8f N1: the blocks of the C host's loader, in all its modes (serial with 1 and 4
    threads, MPI with 1 and 3 threads per rank), against the blocks the REFERENCE's own MPI_csr_load_matrix_block builds
    from the same file (oracle/_ref/ref_dump, the real src/matrix.c:268-419 compiled by oracle/Makefile) -- row pointers,
    columns in stored order, values bit for bit, partition arrays. The file is in shuffled order with 17-digit values,
    short values, exponents and a comment block, on an irregular matrix."""
    A = synth.fem_like(4000, seed=11, scale_decades=1.5)
    row, col, val = A.to_coo()
    rng = np.random.default_rng(8)
    perm = rng.permutation(len(val))
    row, col, val = row[perm], col[perm], val[perm]
    val = np.where(rng.random(len(val)) < 0.2, np.round(val, 3), val)          # some short spellings
    mtx = str(tmp_path / "m.mtx")
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% written by tests/test_host_loader.py\n%\n")
        f.write(f"{A.rows} {A.cols} {len(val)}\n")
        for k, (i, j, v) in enumerate(zip(row.tolist(), col.tolist(), val.tolist())):
            f.write(f"{i + 1} {j + 1} {v!r}\n" if k % 3 else f"{i + 1}  {j + 1}\t{v:.16e}\n")
    ref = str(tmp_path / "ref")
    done = subprocess.run([MPIEXEC, "-n", str(world), REF_DUMP, mtx, "blocks", ref], capture_output=True, text=True, timeout=300)
    if done.returncode != 0 and "unknown method" in done.stderr:
        pytest.skip("oracle/_ref/ref_dump predates the 'blocks' mode: rebuild with make -C oracle")
    assert done.returncode == 0, done.stderr[-400:]
    for mode, threads in (("serial", "1"), ("serial", "4"), ("mpi", "1"), ("mpi", "3")):
        if mode == "mpi" and world == 1:
            continue
        out = str(tmp_path / f"ours_{mode}_{threads}")
        subprocess.run([MPIEXEC, "-n", str(world), DUMP, mtx, out, mode], check=True, timeout=300,
                       env=dict(os.environ, BICG_MTX_THREADS=threads))
        for rank in range(world):
            want = open(f"{ref}.rank{rank}.bin", "rb").read()
            got = open(f"{out}.rank{rank}.bin", "rb").read()
            assert got == want, (mode, threads, rank)
