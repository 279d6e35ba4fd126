"""ctypes binding of oracle/_ref/libref.so -- the REAL reference (solver.c, matrix.c, vector.c,
mmio.c compiled from /root/reference by oracle/Makefile), single MPI rank (MPICH singleton init).

Used only to pin the oracle restatement and to generate tests/golden/ fixtures.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
LIBREF = os.path.join(REF_DIR, "libref.so")
MPI_LIB = "/opt/conda/lib/libmpi.so"
MPIEXEC = "/opt/conda/bin/mpiexec"

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint)


class CSRMatrix(C.Structure):  # reference src/matrix.h:19-26
    _fields_ = [("val", _dp), ("col", _up), ("ptr", _up), ("nz", C.c_uint), ("rows", C.c_uint),
                ("cols", C.c_uint)]


class InfoMatrix(C.Structure):  # reference src/matrix.h:28-33
    _fields_ = [("nz", C.c_uint), ("rows", C.c_uint), ("cols", C.c_uint), ("code", C.c_char * 4),
                ("recvcounts", C.POINTER(C.c_int)), ("displs", C.POINTER(C.c_int))]


def available() -> bool:
    return os.path.exists(LIBREF) and os.path.exists(MPI_LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        mpi = C.CDLL(MPI_LIB, mode=C.RTLD_GLOBAL)
        flag = C.c_int(0)
        mpi.MPI_Initialized(C.byref(flag))
        if not flag.value:
            mpi.MPI_Init(None, None)
        _lib = C.CDLL(LIBREF)
        _lib.my_ddot.restype = C.c_double
        for name in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_int
    return _lib


LIBREF_SHIFTED = os.path.join(REF_DIR, "libref_shifted.so")
_shifted = None


def shifted_lib():
    """reference src/shifted_solver.c (EPS 1e-12, MAX_ITER 1000 compiled in, :5-6)"""
    global _shifted
    if _shifted is None:
        lib()                                    # MPI singleton init
        _shifted = C.CDLL(LIBREF_SHIFTED)
    return _shifted


def solve_shifted(fname, M: RefMatrix, b, sigma, seed):
    """x_set [nsig][n] shift-major (reference src/shifted_solver.c:118-123), r: seed residual."""
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    x = np.zeros(len(sigma) * M.n)
    r = np.array(b, dtype=np.float64)
    fn = getattr(shifted_lib(), fname)
    fn.restype = C.c_int
    args = [C.byref(M.diag), C.byref(M.offd), C.byref(M.info), x.ctypes.data_as(_dp), r.ctypes.data_as(_dp),
            sigma.ctypes.data_as(_dp), C.c_int(len(sigma))]
    if fname != "shifted_bicgstab":
        args.append(C.c_int(seed))
    k = fn(*args)
    return dict(k=k, x=x.reshape(len(sigma), M.n), r=r)


LIBREF_SWITCHING = os.path.join(REF_DIR, "libref_switching.so")
_switching = None


def switching_lib():
    """reference src/shifted_switching_solver.c (EPS 1e-12, MAX_ITER 1000 compiled in, :5-6)"""
    global _switching
    if _switching is None:
        lib()                                    # MPI singleton init
        _switching = C.CDLL(LIBREF_SWITCHING)
    return _switching


def solve_switching(fname, M: "RefMatrix", b, sigma, seed):
    """shifted_lopbicg / shifted_lopbicg_switching / shifted_lopbicg_switching_noovlp; returns k as the
    reference does (the switching variants count from 1)."""
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    x = np.zeros(len(sigma) * M.n)
    r = np.array(b, dtype=np.float64)
    fn = getattr(switching_lib(), fname)
    fn.restype = C.c_int
    k = fn(C.byref(M.diag), C.byref(M.offd), C.byref(M.info), x.ctypes.data_as(_dp), r.ctypes.data_as(_dp),
           sigma.ctypes.data_as(_dp), C.c_int(len(sigma)), C.c_int(seed))
    return dict(k=k, x=x.reshape(len(sigma), M.n), r=r)


class RefMatrix:
    """Single-rank blocks: diag = whole matrix, offd empty (what the loader yields at P = 1)."""

    def __init__(self, A):
        self.keep = (np.ascontiguousarray(A.val, dtype=np.float64), np.ascontiguousarray(A.col, dtype=np.uint32),
                     np.ascontiguousarray(A.ptr, dtype=np.uint32), np.zeros(A.rows + 1, dtype=np.uint32),
                     np.zeros(1), np.zeros(1, dtype=np.uint32),
                     np.array([A.rows], dtype=np.int32), np.array([0], dtype=np.int32))
        v, c, p, zp, zv, zc, cnt, dsp = self.keep
        self.diag = CSRMatrix(v.ctypes.data_as(_dp), c.ctypes.data_as(_up), p.ctypes.data_as(_up), A.nnz, A.rows, A.rows)
        self.offd = CSRMatrix(zv.ctypes.data_as(_dp), zc.ctypes.data_as(_up), zp.ctypes.data_as(_up), 0, A.rows, A.rows)
        self.info = InfoMatrix(A.nnz, A.rows, A.rows, b"MCRG", cnt.ctypes.data_as(C.POINTER(C.c_int)),
                               dsp.ctypes.data_as(C.POINTER(C.c_int)))
        self.n = A.rows


def spmv(M: RefMatrix, x):
    """MPI_csr_spmv_ovlap at one rank (reference src/matrix.c:428-441)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    full = np.zeros(M.n)
    y = np.zeros(M.n)
    lib().MPI_csr_spmv_ovlap(C.byref(M.diag), C.byref(M.offd), C.byref(M.info), x.ctypes.data_as(_dp),
                             full.ctypes.data_as(_dp), y.ctypes.data_as(_dp))
    return y


def ddot(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    return lib().my_ddot(C.c_int(len(x)), x.ctypes.data_as(_dp), y.ctypes.data_as(_dp))


def solve(method, M: RefMatrix, b, krr=0, nrr=0):
    """Calls the reference solver (EPS 1e-15, MAX_ITER 1000 are compiled in, src/solver.c:3-4)."""
    x = np.zeros(M.n)
    r = np.array(b, dtype=np.float64)
    args = [C.byref(M.diag), C.byref(M.offd), C.byref(M.info), x.ctypes.data_as(_dp), r.ctypes.data_as(_dp)]
    if method == "pipe_bicgstab_rr":
        args += [C.c_int(krr), C.c_int(nrr)]
    k = getattr(lib(), method)(*args)
    return dict(k=k, x=x, r=r)
