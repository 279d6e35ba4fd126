"""The unstructured finite-element matrix that stands in for Transport.mtx (mpi_bicgstab_amd.mesh; reference README.md:32-42: a
3-D FEM matrix, symmetric pattern, unsymmetric values): host-side properties, checked without a GPU on small meshes."""
import numpy as np
import pytest

import oracle_lib as O
from mpi_bicgstab_amd import mesh, synth


@pytest.fixture(scope="module")
def small():
    A, ntets = mesh.fem_matrix(16, workers=1)
    return A, ntets


def test_fem_matrix_is_what_its_head_says(small):
    A, ntets = small
    n = 16 ** 3
    assert A.shape == (n, n) and ntets > 5 * n
    P = (A != 0).astype(np.int8)
    assert (P - P.T).nnz == 0                                   # symmetric pattern
    assert abs(A - A.T).max() > 0.1                             # unsymmetric values (the convection term)
    # stiffness and convection rows sum to zero: A 1 = sigma * lumped mass > 0, and the masses add up to the meshed volume
    rs = np.asarray(A.sum(axis=1)).ravel()
    assert rs.min() > 0 and abs(rs.sum() / 1e-3 - 15.0 ** 3) < 0.06 * 15.0 ** 3
    lens = np.diff(A.indptr)
    assert 13.0 < lens.mean() < 17.5 and lens.min() >= 4 and lens.max() <= 40      # Transport.mtx: 14.66 per row
    assert (A.diagonal() > 0).all()
    assert A.has_sorted_indices


def test_fem_matrix_does_not_depend_on_the_number_of_processes():
    """blocks are a function of m only; a tetrahedron belongs to the block that holds its centroid: one process and four give the
    same matrix bit for bit (m = 32: 2 x 1 blocks)"""
    A1, t1 = mesh.fem_matrix(32, workers=1)
    A4, t4 = mesh.fem_matrix(32, workers=4)
    assert len(mesh.blocks_of(32)) == 2 and t1 == t4 and A1.nnz == A4.nnz
    assert np.array_equal(A1.indptr, A4.indptr) and np.array_equal(A1.indices, A4.indices) and np.array_equal(A1.data, A4.data)
    # ... and the block interface leaves no seam: every interior node has a full ring of tetrahedra (>= 10 neighbours)
    lens = np.diff(A1.indptr).reshape(32, 32, 32)
    assert lens[14:18, 4:-4, 4:-4].min() >= 8


@pytest.mark.parametrize("kind", ["generator", "rcm", "random"])
def test_numberings_are_the_same_operator(small, kind):
    A, _ = small
    perm = mesh.numbering(A, kind)
    assert np.array_equal(np.sort(perm), np.arange(A.shape[0]))
    B = mesh.permute(A, perm)
    assert B.has_sorted_indices and B.nnz == A.nnz
    x = np.cos(np.arange(A.shape[0]))
    np.testing.assert_allclose(B @ x[perm], (A @ x)[perm], rtol=1e-12, atol=1e-12)
    C = mesh.to_csr(B)
    row, col, val = C.to_coo()
    # the oracle's product (reference src/matrix.c:498-516) on the synth.CSR form
    np.testing.assert_allclose(O.spmv(C.rows, row, col, val, x[perm]), (A @ x)[perm], rtol=1e-11, atol=1e-11)
    r = np.repeat(np.arange(C.rows), np.diff(C.ptr.astype(np.int64)))
    bw = int(np.abs(C.col.astype(np.int64) - r).max())
    if kind == "rcm":
        assert bw < 2 * 16 * 16 + 3 * 16        # about one and a half planes
    if kind == "random":
        assert bw > C.rows // 2


def test_slab_of_rows_and_scaling(small):
    A, _ = small
    full = mesh.to_csr(A, scale_decades=2.0)
    slab = mesh.to_csr(A, scale_decades=2.0, rows=(1000, 2200))
    p = full.ptr.astype(np.int64)
    assert slab.rows == 1200 and slab.cols == full.rows
    assert np.array_equal(slab.col, full.col[p[1000]:p[2200]]) and np.array_equal(slab.val, full.val[p[1000]:p[2200]])
    d = synth.row_scale(np.arange(full.rows), 2.0)
    plain = mesh.to_csr(A)
    rid = np.repeat(np.arange(full.rows), np.diff(p))
    np.testing.assert_array_equal(full.val, plain.val * d[rid] * d[plain.col.astype(np.int64)])


def test_cache_files_give_the_same_matrix(tmp_path):
    a = mesh.fem_unstructured(12, "rcm", scale_decades=1.0, cache_dir=str(tmp_path), workers=1)
    assert len(list(tmp_path.glob("bicg_mesh_m12_*.npy"))) == 3
    mesh._CACHE.clear()
    b = mesh.fem_unstructured(12, "rcm", scale_decades=1.0, cache_dir=str(tmp_path), workers=1)      # read back
    mesh._CACHE.clear()
    c = mesh.fem_unstructured(12, "rcm", scale_decades=1.0, workers=1)                                  # regenerated
    for other in (b, c):
        assert np.array_equal(a.ptr, other.ptr) and np.array_equal(a.col, other.col) and np.array_equal(a.val, other.val)


def test_window_statistics_tell_the_numberings_apart():
    """what the LDS window of the ragged-rows product would have to hold per 256-row group: the generator order touches a few long
    runs of consecutive columns, reverse Cuthill-McKee many short ones, a random permutation one column per run"""
    stats = {}
    for kind in ("generator", "rcm", "random"):
        C = mesh.fem_unstructured(24, kind, workers=1)
        distinct, runs = mesh.window_stats(C)
        stats[kind] = (np.median(distinct), np.median(runs))
    assert stats["random"][1] > 0.6 * stats["random"][0] and stats["random"][0] > 2.0 * stats["generator"][0]       # (13 824 columns: some land side by side)
    assert stats["rcm"][1] > stats["generator"][1]
