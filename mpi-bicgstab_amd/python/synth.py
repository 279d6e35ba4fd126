"""Synthetic CSR matrices and the reference's row-block partition (host-side, numpy only).

The reference reads its matrices from Matrix-Market files (reference src/matrix.c:268-419) and the
real Transport.mtx is not available offline, so benchmarks and tests use the generators below
(SURVEY.md section 8d "Synthetic inputs"):

* ``transport_like``  -- Transport-*shaped*: n = 1 602 111 rows, 15 fixed diagonals
  {0, +-1, +-117, +-118, +-13689, +-13690, +-13806, +-13807} clipped to [0, n)
  => 23 921 209 non-zeros (Transport.mtx itself: 23 487 281, reference README.md:32-42).
* ``banded``          -- dense band of half-bandwidth b.
* ``stencil7``        -- 7-point stencil on an m^3 grid (tiny KAT: m = 12, nonsymmetric weights).

Right-hand sides follow reference src/main.c:109-117: b = A * 1, x0 = 0.

``split_blocks`` mirrors MPI_coo_load_matrix_block's partition and diag/offd split
(reference src/matrix.c:295-308, 336-340, 380-392) for in-memory CSR input.
"""
from __future__ import annotations

import dataclasses

import numpy as np

TRANSPORT_N = 1_602_111
TRANSPORT_OFFSETS = (0, 1, -1, 117, -117, 118, -118, 13689, -13689, 13690, -13690, 13806, -13806,
                     13807, -13807)


@dataclasses.dataclass
class CSR:
    """CSR with the reference's index width (uint32, reference src/matrix.h:19-26)."""
    rows: int
    cols: int
    ptr: np.ndarray   # uint32 [rows+1]
    col: np.ndarray   # uint32 [nnz]
    val: np.ndarray   # float64 [nnz]

    @property
    def nnz(self) -> int:
        return int(self.ptr[-1])

    def matvec(self, x: np.ndarray) -> np.ndarray:
        """y = A x with numpy (pairwise sums; NOT the oracle -- only for building b = A*1)."""
        prod = self.val * x[self.col]
        ptr = self.ptr.astype(np.int64)
        y = np.zeros(self.rows)
        nonempty = ptr[1:] > ptr[:-1]
        y[nonempty] = np.add.reduceat(prod, ptr[:-1][nonempty])
        return y

    def to_coo(self):
        row = np.repeat(np.arange(self.rows, dtype=np.uint32), np.diff(self.ptr.astype(np.int64)))
        return row, self.col.copy(), self.val.copy()


def _uniform(idx: np.ndarray, seed: int) -> np.ndarray:
    """Counter-based U[0,1): splitmix64 of (seed + idx), top 53 bits."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def row_scale(idx: np.ndarray, decades: float, seed: int = 99) -> np.ndarray:
    """d_i = 10^(decades * (U_i - 0.5)): the symmetric diagonal scaling D A D used to make the
    synthetic systems as badly scaled as real FEM matrices (Transport.mtx needs ~2700 BiCGStab
    iterations, reference README.md:44-45; the unscaled synthetic converges in ~20)."""
    return 10.0 ** (decades * (_uniform(idx.astype(np.int64) + 7_000_000_000, seed) - 0.5))


def from_offsets(n: int, offsets, diag_base: float = 16.0, seed: int = 12345, rows=None,
                 scale_decades: float = 0.0) -> CSR:
    """Rows hold one entry per offset (clipped to [0, n)), ascending column order.

    diagonal = diag_base + U[0,1); off-diagonal = -(0.5 + 0.5 U[0,1)).  diag_base = 16 is the
    strongly dominant law of SURVEY.md section 8d; smaller values give harder systems.
    rows=(lo, hi) builds only that row range of the same global matrix (GLOBAL columns, cols = n).
    scale_decades > 0 applies A <- D A D with D = row_scale (harder, same sparsity pattern).
    """
    offs = np.array(sorted(offsets), dtype=np.int64)
    k = len(offs)
    lo, hi = (0, n) if rows is None else rows
    rows = np.arange(lo, hi, dtype=np.int64)
    cols = rows[:, None] + offs[None, :]                     # [n, k]
    ok = (cols >= 0) & (cols < n)
    counts = ok.sum(axis=1)
    ptr = np.zeros(hi - lo + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    eid = rows[:, None] * k + np.arange(k, dtype=np.int64)[None, :]
    u = _uniform(eid[ok], seed)
    is_diag = np.broadcast_to(offs[None, :] == 0, cols.shape)[ok]
    val = np.where(is_diag, diag_base + u, -(0.5 + 0.5 * u))
    if scale_decades > 0.0:
        rid = np.broadcast_to(rows[:, None], cols.shape)[ok]
        val = val * row_scale(rid, scale_decades) * row_scale(cols[ok], scale_decades)
    return CSR(hi - lo, n, ptr.astype(np.uint32), cols[ok].astype(np.uint32), val)


def transport_like(n: int = TRANSPORT_N, diag_base: float = 16.0, seed: int = 12345, rows=None,
                   scale_decades: float = 0.0) -> CSR:
    return from_offsets(n, TRANSPORT_OFFSETS, diag_base, seed, rows, scale_decades)


def fem_like(n: int = TRANSPORT_N, seed: int = 4242, keep: float = 0.55, rows=None, scale_decades: float = 0.0) -> CSR:
    """Irregular rows like an unstructured FEM matrix: a 27-offset 3-D stencil (117 x 117 x ~117
    node numbering) from which every off-diagonal entry is kept with probability `keep`, so row
    lengths vary between ~6 and 27 (mean ~15.3 at keep = 0.55; Transport.mtx: 14.66). Used to
    exercise the sliced-ELL / CSR hybrid on ragged rows; values follow from_offsets' law.
    rows=(lo, hi): only that row range of the same global matrix (GLOBAL columns); scale_decades as in
    from_offsets (a harder system: bench.py needs a few hundred unconverged iterations)."""
    nx = 117
    offs = np.array(sorted({dz * nx * nx + dy * nx + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)}),
                    dtype=np.int64)
    A = from_offsets(n, offs.tolist(), diag_base=32.0, seed=seed, rows=rows, scale_decades=scale_decades)
    lo = 0 if rows is None else rows[0]
    ptr = A.ptr.astype(np.int64)
    rowid = np.repeat(np.arange(lo, lo + A.rows, dtype=np.int64), np.diff(ptr))
    col = A.col.astype(np.int64)
    is_diag = col == rowid
    # the decision depends on (row, offset) only, so every slab sees the same global matrix
    eid = rowid * len(offs) + np.searchsorted(offs, col - rowid)
    keep_mask = is_diag | (_uniform(eid, seed + 1) < keep)
    cnt = np.bincount(rowid[keep_mask] - lo, minlength=A.rows)
    p2 = np.zeros(A.rows + 1, dtype=np.int64)
    np.cumsum(cnt, out=p2[1:])
    return CSR(A.rows, n, p2.astype(np.uint32), A.col[keep_mask], A.val[keep_mask])


def transport_nnz(n: int = TRANSPORT_N) -> int:
    """non-zeros of transport_like(n) without building it"""
    return sum(max(n - abs(o), 0) for o in TRANSPORT_OFFSETS)


def split_row_slab(slab: CSR, lo: int):
    """diag/offd blocks of a row slab that starts at global row `lo` (slab has GLOBAL columns):
    what MPI_coo_load_matrix_block gives one rank (reference src/matrix.c:336-340, 380-392)."""
    hi = lo + slab.rows
    ptr = slab.ptr.astype(np.int64)
    col = slab.col.astype(np.int64)
    rowid = np.repeat(np.arange(slab.rows), np.diff(ptr))
    local = (col >= lo) & (col < hi)

    def build(mask, ncols, shift):
        p = np.zeros(slab.rows + 1, dtype=np.int64)
        np.cumsum(np.bincount(rowid[mask], minlength=slab.rows), out=p[1:])
        return CSR(slab.rows, ncols, p.astype(np.uint32), (col[mask] - shift).astype(np.uint32), slab.val[mask].copy())

    return build(local, slab.rows, lo), build(~local, slab.cols, 0)


def banded(n: int, half_bw: int, diag_base: float | None = None, seed: int = 777, rows=None, scale_decades: float = 0.0) -> CSR:
    """dense band of half-bandwidth b (SURVEY.md section 8d synthetic input (ii)); rows=(lo, hi): a row slab"""
    if diag_base is None:
        diag_base = 2.0 * half_bw + 1.0
    return from_offsets(n, range(-half_bw, half_bw + 1), diag_base, seed, rows, scale_decades)


def banded_nnz(n: int, half_bw: int) -> int:
    return sum(max(n - abs(o), 0) for o in range(-half_bw, half_bw + 1))


def banded_rows_for(nnz_target: int, half_bw: int) -> int:
    """rows of the band matrix with about nnz_target non-zeros"""
    return max(2 * half_bw + 2, int(round(nnz_target / (2 * half_bw + 1))))


LAPLACE_WEIGHTS = (6.0, -1.0, -1.0, -1.0, -1.0, -1.0, -1.0)


def stencil7(m: int, weights=(6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.0), rows=None) -> CSR:
    """7-point stencil on an m^3 grid; weights = (centre, x-, x+, y-, y+, z-, z+).
    rows=(lo, hi) builds only that row range (GLOBAL columns, cols = m^3): one GPU's slab of
    BASELINE.json configs[3] (512^3 Laplacian: 8 slabs of 64 planes)."""
    n = m ** 3
    lo, hi = (0, n) if rows is None else rows
    idx = np.arange(lo, hi, dtype=np.int64)
    ix, iy, iz = idx % m, (idx // m) % m, idx // (m * m)
    cand = [  # (offset, mask, weight) in ascending column order
        (-m * m, iz > 0, weights[5]), (-m, iy > 0, weights[3]), (-1, ix > 0, weights[1]),
        (0, np.ones(hi - lo, bool), weights[0]),
        (1, ix < m - 1, weights[2]), (m, iy < m - 1, weights[4]), (m * m, iz < m - 1, weights[6]),
    ]
    cols = np.stack([idx + o for o, _, _ in cand], axis=1)
    ok = np.stack([msk for _, msk, _ in cand], axis=1)
    vals = np.broadcast_to(np.array([w for _, _, w in cand], dtype=np.float64)[None, :], cols.shape)
    ptr = np.zeros(hi - lo + 1, dtype=np.int64)
    np.cumsum(ok.sum(axis=1), out=ptr[1:])
    return CSR(hi - lo, n, ptr.astype(np.uint32), cols[ok].astype(np.uint32), vals[ok].copy())


def grid7(nx: int, ny: int, nz: int, weights=(6.5, -1.2, -0.8, -1.1, -0.9, -1.0, -1.0), upper_weights=None, wrap_y: bool = False) -> CSR:
    """7-point stencil on an nx x ny x nz grid (x fastest), weights as in stencil7. upper_weights: the coefficients of the planes
    z >= nz // 2 (two value lists for one distance list). wrap_y: the rows of a plane's first / last line keep their -nx / +nx
    entry where that row exists (it lies in the neighbouring plane) -- same distances, different rows have them."""
    n = nx * ny * nz
    idx = np.arange(n, dtype=np.int64)
    ix, iy, iz = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    ylo = (idx - nx >= 0) if wrap_y else (iy > 0)
    yhi = (idx + nx < n) if wrap_y else (iy < ny - 1)
    order = (5, 3, 1, 0, 2, 4, 6)          # weights' positions in ascending column order
    masks = (iz > 0, ylo, ix > 0, np.ones(n, bool), ix < nx - 1, yhi, iz < nz - 1)
    offs = (-nx * ny, -nx, -1, 0, 1, nx, nx * ny)
    cols = np.stack([idx + o for o in offs], axis=1)
    ok = np.stack(masks, axis=1)
    w_lo = np.array([weights[k] for k in order], dtype=np.float64)
    vals = np.broadcast_to(w_lo[None, :], cols.shape).copy()
    if upper_weights is not None:
        vals[iz >= nz // 2, :] = np.array([upper_weights[k] for k in order], dtype=np.float64)[None, :]
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(ok.sum(axis=1), out=ptr[1:])
    return CSR(n, n, ptr.astype(np.uint32), cols[ok].astype(np.uint32), vals[ok].copy())


def stencil7_nnz(m: int) -> int:
    return 7 * m ** 3 - 6 * m * m


def stencil7_matvec(m: int, weights, x: np.ndarray) -> np.ndarray:
    """y = A x for stencil7(m, weights) WITHOUT the matrix: every row's products added in stored (ascending column) order,
    one rounding per product and per sum -- the arithmetic of the reference's mult() (src/matrix.c:506-515) without FMA,
    so the result is bit-identical to it (a missing neighbour adds nothing; 0 + p = p). Grid-sized temporaries only:
    this is how the 512^3 SpMV (134 M rows; no CPU oracle run fits a test) is checked."""
    X = np.ascontiguousarray(x, dtype=np.float64).reshape(m, m, m)      # [z][y][x]
    Y = np.zeros_like(X)
    Y[1:, :, :] += weights[5] * X[:-1, :, :]        # z-   (column - m^2)
    Y[:, 1:, :] += weights[3] * X[:, :-1, :]        # y-   (column - m)
    Y[:, :, 1:] += weights[1] * X[:, :, :-1]        # x-   (column - 1)
    Y += weights[0] * X                             # centre
    Y[:, :, :-1] += weights[2] * X[:, :, 1:]        # x+
    Y[:, :-1, :] += weights[4] * X[:, 1:, :]        # y+
    Y[:-1, :, :] += weights[6] * X[1:, :, :]        # z+
    return Y.reshape(-1)


def random_rows(n: int, max_row: int, seed: int = 1, empty_frac: float = 0.1, long_rows=()) -> CSR:
    """Ragged test matrix: random row lengths in [0, max_row], some empty rows, optional very long
    rows (row index -> length); diagonal made dominant where present."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, max_row + 1, size=n)
    lens[rng.random(n) < empty_frac] = 0
    for r, ln in dict(long_rows).items():
        lens[r] = min(ln, n)
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    col = np.empty(int(ptr[-1]), dtype=np.uint32)
    val = rng.uniform(-1.0, 1.0, size=int(ptr[-1]))
    for i in range(n):
        ln = int(lens[i])
        if ln:
            c = rng.choice(n, size=ln, replace=False)
            if ln < 64:
                c[0] = i                      # keep a diagonal entry in short rows
            c.sort()
            col[ptr[i]:ptr[i + 1]] = c
            seg = val[ptr[i]:ptr[i + 1]]
            d = np.nonzero(c == i)[0]
            if d.size:
                seg[d[0]] = np.abs(seg).sum() + 1.0
    return CSR(n, n, ptr.astype(np.uint32), col, val)


def colmajor_coo(A: CSR):
    """COO triplets in column-major order (how SuiteSparse .mtx files are written)."""
    row, col, val = A.to_coo()
    order = np.lexsort((row, col))
    return row[order], col[order], val[order]


def write_mtx(path: str, A: CSR, order: str = "colmajor") -> None:
    """Write 'coordinate real general' (the only flavour the reference block loader handles,
    SURVEY.md section 4 defect 2); 17 significant digits so values round-trip exactly."""
    if order == "colmajor":
        row, col, val = colmajor_coo(A)
    else:
        row, col, val = A.to_coo()
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{A.rows} {A.cols} {A.nnz}\n")
        for i, j, v in zip(row.tolist(), col.tolist(), val.tolist()):
            f.write(f"{i + 1} {j + 1} {v!r}\n")


def partition(n: int, nranks: int):
    """(counts, displs) of reference src/matrix.c:295-308."""
    base, extra = divmod(n, nranks)
    counts = np.array([base + (1 if p < extra else 0) for p in range(nranks)], dtype=np.int32)
    displs = np.zeros(nranks, dtype=np.int32)
    np.cumsum(counts[:-1], out=displs[1:])
    return counts, displs


def partition_nnz(A: CSR, nranks: int):
    """non-zero balanced contiguous row blocks through the library's bicg_partition_nnz"""
    import ctypes as C
    from . import hipsolver as H
    lens = np.ascontiguousarray(np.diff(A.ptr.astype(np.int64)), dtype=np.uint32)
    counts = np.zeros(nranks, dtype=np.int32)
    displs = np.zeros(nranks, dtype=np.int32)
    ip = C.POINTER(C.c_int)
    H.lib().bicg_partition_nnz(lens.ctypes.data_as(C.POINTER(C.c_uint)), C.c_uint(A.rows), C.c_int(nranks),
                               counts.ctypes.data_as(ip), displs.ctypes.data_as(ip))
    return counts, displs


def split_blocks(A: CSR, nranks: int, rank: int, part=None):
    """(diag, offd, counts, displs) for one rank: diag has LOCAL columns and cols = local rows,
    offd keeps GLOBAL columns and cols = n (reference src/matrix.c:343-351, 380-392).
    part = (counts, displs) overrides the reference's equal-rows partition."""
    counts, displs = part if part is not None else partition(A.rows, nranks)
    lo, hi = int(displs[rank]), int(displs[rank] + counts[rank])
    ptr = A.ptr.astype(np.int64)
    a, b = int(ptr[lo]), int(ptr[hi])
    col = A.col[a:b].astype(np.int64)
    val = A.val[a:b]
    rowid = np.repeat(np.arange(hi - lo), np.diff(ptr[lo:hi + 1]))
    local = (col >= lo) & (col < hi)

    def build(mask, ncols, shift):
        cnt = np.bincount(rowid[mask], minlength=hi - lo)
        p = np.zeros(hi - lo + 1, dtype=np.int64)
        np.cumsum(cnt, out=p[1:])
        return CSR(hi - lo, ncols, p.astype(np.uint32), (col[mask] - shift).astype(np.uint32),
                   val[mask].copy())

    return build(local, hi - lo, lo), build(~local, A.cols, 0), counts, displs
