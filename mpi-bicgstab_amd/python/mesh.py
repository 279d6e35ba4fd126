"""An UNSTRUCTURED finite-element matrix as the stand-in for Transport.mtx (host-side, numpy + scipy only).

Transport.mtx (reference README.md:32-42) is a 3-D finite-element matrix: 1 602 111 rows, 23 487 281 non-zeros, symmetric pattern,
unsymmetric values. The file is not available offline; the generators of synth.py number their nodes like a grid, so every 256-row
group of theirs touches a handful of runs of consecutive columns. This module builds what such a file holds when nothing is
regular: the P1 (linear tetrahedra) discretisation of

        -eps Laplace(u) + beta . grad(u) + sigma u

on a Delaunay tetrahedralisation of m^3 jittered points (m = 117: 1 601 613 nodes, ~16 non-zeros per row), in three numberings:

  "generator"  the order the points were made in (x fastest) -- what a mesh generator that sweeps a box writes out;
  "rcm"        reverse Cuthill-McKee of the assembled graph (scipy.sparse.csgraph) -- what a bandwidth-reducing pre-pass gives;
  "random"     a random permutation -- the adversarial case: no two neighbours are near each other in memory.

The tetrahedralisation is made block by block (scipy.spatial.Delaunay = Qhull on a block of (z, y) point lines + a margin of 4
lines on every side, one process per block) and is independent of the number of processes: a tetrahedron is kept by the block that
holds its centroid, and only when its circumradius is <= 1.5 grid spacings -- every ball of radius >= 0.87 + 1.74 * jitter holds a
point, so no interior Delaunay tetrahedron is larger, and one that small has its whole circumsphere inside block + margin, where the
block's triangulation and the global one agree (the flat hull tetrahedra with huge circumspheres are what is dropped).

Element matrices (per tetrahedron T with volume V, vertex gradients g_i of the hat functions):
  K_ij = eps V g_i.g_j          C_ij = (V / 4) beta(c_T).g_j          M_ii = sigma V / 4 (lumped)
beta is a rotating field around the box's axis (divergence-free, so the rows of K + C sum to zero and A 1 = sigma M 1 > 0).
"""
from __future__ import annotations

import os

import numpy as np

from .synth import CSR, row_scale

MARGIN = 4
RMAX = 1.5


def points(m: int, jitter: float = 0.3, seed: int = 2024, box=None):
    """-> (global node numbers, coordinates [n, 3] = x, y, z) of the points with z in [z0, z1) and y in [y0, y1) (box = (z0, z1, y0,
    y1), default: all), in generator order (x fastest, then y, then z)"""
    from .synth import _uniform
    z0, z1, y0, y1 = (0, m, 0, m) if box is None else box
    zz, yy, xx = np.meshgrid(np.arange(z0, z1, dtype=np.int64), np.arange(y0, y1, dtype=np.int64), np.arange(m, dtype=np.int64), indexing="ij")
    idx = ((zz * m + yy) * m + xx).ravel()
    g = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], axis=1).astype(np.float64)
    u = np.stack([_uniform(3 * idx + k, seed) for k in range(3)], axis=1)
    return idx, g + jitter * (2.0 * u - 1.0)


def _circumradius2(e1, e2, e3):
    """squared circumradius and signed 6 x volume of the tetrahedra (p0, p0 + e1, p0 + e2, p0 + e3)"""
    c23, c31, c12 = np.cross(e2, e3), np.cross(e3, e1), np.cross(e1, e2)
    det = np.einsum("ij,ij->i", e1, c23)
    l1, l2, l3 = (np.einsum("ij,ij->i", e, e) for e in (e1, e2, e3))
    with np.errstate(divide="ignore", invalid="ignore"):
        c = (l1[:, None] * c23 + l2[:, None] * c31 + l3[:, None] * c12) / (2.0 * det[:, None])
    r2 = np.einsum("ij,ij->i", c, c)
    return np.where(np.isfinite(r2), r2, np.inf), det


def block_tets(m: int, box, jitter: float, seed: int):
    """-> (tets [t, 4] as positions in the block's point list, that list's global node numbers, its coordinates): the kept
    tetrahedra whose centroid lies in the block box = (z0, z1, y0, y1); the point list is block + margin"""
    from scipy.spatial import Delaunay
    z0, z1, y0, y1 = box
    ext = (max(0, z0 - MARGIN), min(m, z1 + MARGIN), max(0, y0 - MARGIN), min(m, y1 + MARGIN))
    idx, p = points(m, jitter, seed, ext)
    tri = Delaunay(p).simplices
    q = p[tri]                                              # [t, 4, 3]
    cen = q.mean(axis=1)
    r2, det = _circumradius2(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0], q[:, 3] - q[:, 0])
    # the nodes of line k sit around coordinate k: the block [a, b) owns [a - 0.5, b - 0.5), the outermost blocks everything beyond
    lo = lambda a: -np.inf if a == 0 else a - 0.5
    hi = lambda b: np.inf if b == m else b - 0.5
    keep = ((cen[:, 2] >= lo(z0)) & (cen[:, 2] < hi(z1)) & (cen[:, 1] >= lo(y0)) & (cen[:, 1] < hi(y1)) &
            (r2 <= RMAX * RMAX) & (np.abs(det) > 1e-9))
    tri = tri[keep].astype(np.int64)
    neg = det[keep] < 0                                     # positive orientation: swap two vertices where it is not
    tri[neg, 0], tri[neg, 1] = tri[neg, 1].copy(), tri[neg, 0].copy()
    return tri, idx, p


def assemble(n: int, tets: np.ndarray, idx: np.ndarray, p: np.ndarray, eps: float, beta: float, sigma: float, centre):
    """P1 element matrices of `tets` (positions in the point list p, global numbers idx) summed into a scipy CSR [n, n]"""
    import scipy.sparse as sp
    q = p[tets]                                             # [t, 4, 3]
    e1, e2, e3 = q[:, 1] - q[:, 0], q[:, 2] - q[:, 0], q[:, 3] - q[:, 0]
    c23, c31, c12 = np.cross(e2, e3), np.cross(e3, e1), np.cross(e1, e2)
    det = np.einsum("ij,ij->i", e1, c23)
    vol = det / 6.0
    g = np.empty((len(tets), 4, 3))
    g[:, 1], g[:, 2], g[:, 3] = c23 / det[:, None], c31 / det[:, None], c12 / det[:, None]
    g[:, 0] = -(g[:, 1] + g[:, 2] + g[:, 3])
    cen = q.mean(axis=1)
    # rotation around the axis through `centre` parallel to z, plus a constant drift along z: divergence-free
    bx = -beta * (cen[:, 1] - centre[1]) / centre[1]
    by = beta * (cen[:, 0] - centre[0]) / centre[0]
    bz = np.full(len(tets), 0.5 * beta)
    bg = bx[:, None] * g[:, :, 0] + by[:, None] * g[:, :, 1] + bz[:, None] * g[:, :, 2]      # beta . g_j   [t, 4]
    E = eps * vol[:, None, None] * np.einsum("tik,tjk->tij", g, g)
    E += (vol / 4.0)[:, None, None] * bg[:, None, :]
    E[:, np.arange(4), np.arange(4)] += (sigma * vol / 4.0)[:, None]
    gt = idx[tets].astype(np.int32)
    I = np.broadcast_to(gt[:, :, None], E.shape)
    J = np.broadcast_to(gt[:, None, :], E.shape)
    A = sp.coo_matrix((E.ravel(), (I.ravel(), J.ravel())), shape=(n, n)).tocsr()     # (duplicates added in scipy's order)
    A.sort_indices()
    return A


def _block_job(args):
    m, box, jitter, seed, eps, beta, sigma = args
    tets, idx, p = block_tets(m, box, jitter, seed)
    A = assemble(m ** 3, tets, idx, p, eps, beta, sigma, ((m - 1) / 2.0, (m - 1) / 2.0))
    # only the rows the block touches travel back to the parent
    rows = np.flatnonzero(np.diff(A.indptr))
    return rows.astype(np.int64), np.diff(A.indptr)[rows].astype(np.int64), A.indices, A.data, len(tets)


def blocks_of(m: int):
    """the fixed decomposition (a function of m only, so the matrix does not depend on the machine): blocks of about 15 planes
    x 30 lines (m = 117: 8 x 4 = 32 blocks)"""
    def cuts(step):
        k = max(1, round(m / step))
        return [(i * m) // k for i in range(k)] + [m]
    cz, cy = cuts(15), cuts(30)
    return [(cz[i], cz[i + 1], cy[j], cy[j + 1]) for i in range(len(cz) - 1) for j in range(len(cy) - 1)]


def fem_matrix(m: int = 117, jitter: float = 0.3, seed: int = 2024, eps: float = 1.0, beta: float = 1.0, sigma: float = 1e-3,
               workers: int | None = None):
    """-> (scipy CSR in generator order, number of tetrahedra)"""
    import scipy.sparse as sp
    jobs = [(m, box, jitter, seed, eps, beta, sigma) for box in blocks_of(m)]
    workers = min(len(jobs), workers if workers else len(os.sched_getaffinity(0)))
    if workers > 1 and m >= 24:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_block_job, jobs, chunksize=1)
    else:
        parts = [_block_job(j) for j in jobs]
    n = m ** 3
    # one COO -> CSR pass over the blocks' (already row-summed) entries, in the fixed block order: the rows on block interfaces get
    # their two to four contributions added in that order whatever `workers` is
    row = np.concatenate([np.repeat(r, l) for r, l, _, _, _ in parts]).astype(np.int32)
    col = np.concatenate([p[2] for p in parts]).astype(np.int32)
    val = np.concatenate([p[3] for p in parts])
    A = sp.coo_matrix((val, (row, col)), shape=(n, n)).tocsr()
    A.sort_indices()
    return A, sum(p[4] for p in parts)


def numbering(A, kind: str, seed: int = 7) -> np.ndarray:
    """perm[new] = old node"""
    n = A.shape[0]
    if kind == "generator":
        return np.arange(n, dtype=np.int64)
    if kind == "rcm":
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        return reverse_cuthill_mckee(A, symmetric_mode=True).astype(np.int64)
    if kind == "random":
        return np.random.default_rng(seed).permutation(n).astype(np.int64)
    raise ValueError(kind)


def permute(A, perm: np.ndarray):
    """P A P^T with sorted column indices"""
    import scipy.sparse as sp
    n = A.shape[0]
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n, dtype=np.int64)
    ptr = A.indptr.astype(np.int64)
    lens = np.diff(ptr)[perm]
    p2 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=p2[1:])
    # gather the rows in their new order, rename the columns
    src = np.repeat(ptr[perm] - p2[:-1], lens) + np.arange(int(p2[-1]), dtype=np.int64)
    B = sp.csr_matrix((A.data[src], inv[A.indices[src]].astype(np.int32), p2), shape=(n, n))
    B.sort_indices()
    return B


def to_csr(A, scale_decades: float = 0.0, rows=None) -> CSR:
    """synth.CSR (uint32 indices) of a scipy CSR; D A D with synth.row_scale over `scale_decades`; rows=(lo, hi): that slab"""
    n = A.shape[0]
    lo, hi = (0, n) if rows is None else rows
    ptr = A.indptr.astype(np.int64)
    a, b = int(ptr[lo]), int(ptr[hi])
    col = A.indices[a:b].astype(np.int64)
    val = A.data[a:b].astype(np.float64)
    if scale_decades > 0.0:
        rid = np.repeat(np.arange(lo, hi, dtype=np.int64), np.diff(ptr[lo:hi + 1]))
        val = val * row_scale(rid, scale_decades) * row_scale(col, scale_decades)
    return CSR(hi - lo, n, (ptr[lo:hi + 1] - a).astype(np.uint32), col.astype(np.uint32), val)


_CACHE: dict = {}


def fem_unstructured(m: int = 117, numbering_kind: str = "rcm", scale_decades: float = 0.0, rows=None, jitter: float = 0.3, seed: int = 2024,
                     workers: int | None = None, cache_dir: str | None = None) -> CSR:
    """the matrix of this module's head in one of the three numberings. The assembled matrix (generator order) is kept in this
    process, and in `cache_dir` (np.save files; ranks of one host share one generation) when given."""
    key = (m, jitter, seed)
    A = _CACHE.get(key)
    if A is None:
        import scipy.sparse as sp
        files = None
        if cache_dir:
            stem = os.path.join(cache_dir, f"bicg_mesh_m{m}_j{jitter}_s{seed}")
            files = [stem + s for s in ("_ptr.npy", "_ind.npy", "_dat.npy")]
        if files and all(os.path.exists(f) for f in files):
            ptr, ind, dat = (np.load(f, mmap_mode="r") for f in files)
            A = sp.csr_matrix((np.asarray(dat), np.asarray(ind), np.asarray(ptr)), shape=(m ** 3, m ** 3))
        else:
            A, _ = fem_matrix(m, jitter, seed, workers=workers)
            if files:
                for f, arr in zip(files, (A.indptr, A.indices, A.data)):
                    np.save(f + ".tmp.npy", arr)
                    os.replace(f + ".tmp.npy", f)
        _CACHE.clear()
        _CACHE[key] = A
    pk = (key, numbering_kind)
    B = _CACHE.get(pk)
    if B is None:
        B = A if numbering_kind == "generator" else permute(A, numbering(A, numbering_kind))
        for k in [k for k in _CACHE if isinstance(k[0], tuple)]:
            del _CACHE[k]
        _CACHE[pk] = B
    return to_csr(B, scale_decades, rows)


def window_stats(A: CSR, group: int = 256):
    """per `group`-row block of a square matrix: distinct columns touched and runs of consecutive columns among them
    (what the LDS window of the ragged-rows product has to hold) -> (distinct[], runs[])"""
    ptr = A.ptr.astype(np.int64)
    ng = -(-A.rows // group)
    distinct = np.zeros(ng, dtype=np.int64)
    runs = np.zeros(ng, dtype=np.int64)
    for g in range(ng):
        c = np.unique(A.col[ptr[g * group]:ptr[min(A.rows, (g + 1) * group)]].astype(np.int64))
        distinct[g] = c.size
        runs[g] = 1 + int(np.count_nonzero(np.diff(c) != 1)) if c.size else 0
    return distinct, runs
