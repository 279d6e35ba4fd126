"""ctypes binding of oracle/liboracle.so (the CPU oracle; test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
METHODS = {"bicgstab": 0, "ca_bicgstab": 1, "pipe_bicgstab": 2, "pipe_bicgstab_rr": 3}

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint)


class OrcOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int), ("krr", C.c_int), ("nrr", C.c_int),
                ("tr_alpha", _dp), ("tr_omega", _dp), ("tr_beta", _dp), ("tr_dotr", _dp),
                ("dot_zero", C.c_double), ("dot_r", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        _lib = C.CDLL(so)
        _lib.orc_solve_coo.restype = C.c_int
        _lib.orc_solve_coo.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_uint, _up, _up, _dp, _dp, _dp,
                                       C.POINTER(OrcOpts)]
        _lib.orc_spmv_coo.restype = None
        _lib.orc_spmv_coo.argtypes = [C.c_int, C.c_uint, C.c_uint, _up, _up, _dp, _dp, _dp]
        _ip = C.POINTER(C.c_int)
        _lib.orc_solve_coo_part.restype = C.c_int
        _lib.orc_solve_coo_part.argtypes = [C.c_int, C.c_int, _ip, C.c_uint, C.c_uint, _up, _up, _dp, _dp, _dp,
                                            C.POINTER(OrcOpts)]
        _lib.orc_spmv_coo_part.restype = None
        _lib.orc_spmv_coo_part.argtypes = [C.c_int, _ip, C.c_uint, C.c_uint, _up, _up, _dp, _dp, _dp]
        _lib.orc_ddot.restype = C.c_double
        _lib.orc_ddot.argtypes = [C.c_int, _dp, _dp]
        _lib.orc_daxpy.restype = None
        _lib.orc_daxpy.argtypes = [C.c_int, C.c_double, _dp, _dp]
        _lib.orc_dscal.restype = None
        _lib.orc_dscal.argtypes = [C.c_int, C.c_double, _dp]
        _lib.orc_read_mtx.restype = C.c_int
        _lib.orc_shifted_coo.restype = C.c_int
        _lib.orc_shifted_coo.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_uint, _up, _up, _dp, _dp, _dp, _dp, C.c_int, C.c_int,
                                         C.POINTER(OrcOpts)]
        _lib.orc_shifted_lop_coo.restype = C.c_int
        _lib.orc_shifted_lop_coo.argtypes = [C.c_int, C.c_uint, C.c_uint, _up, _up, _dp, _dp, _dp, _dp, C.c_int, C.c_int,
                                             C.POINTER(OrcOpts)]
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _u(a):
    return a.ctypes.data_as(_up)


def _coo(row, col, val):
    return (np.ascontiguousarray(row, dtype=np.uint32), np.ascontiguousarray(col, dtype=np.uint32),
            np.ascontiguousarray(val, dtype=np.float64))


def _counts(counts, nranks):
    c = np.ascontiguousarray(counts, dtype=np.int32)
    assert len(c) == nranks
    return c, c.ctypes.data_as(C.POINTER(C.c_int))


def spmv(n, row, col, val, x, nranks=1, counts=None):
    """y = A x through the oracle's distributed SpMV over `nranks` virtual ranks (file-order COO);
    counts = rows per rank for a non-default contiguous partition."""
    row, col, val = _coo(row, col, val)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(n)
    if counts is not None:
        keep, cp = _counts(counts, nranks)
        lib().orc_spmv_coo_part(nranks, cp, n, len(val), _u(row), _u(col), _d(val), _d(x), _d(y))
    else:
        lib().orc_spmv_coo(nranks, n, len(val), _u(row), _u(col), _d(val), _d(x), _d(y))
    return y


def ddot(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    return lib().orc_ddot(len(x), _d(x), _d(y))


def solve(method, n, row, col, val, b, x0=None, nranks=1, tol=1e-15, max_iter=1000, krr=0, nrr=0, counts=None):
    """Returns dict(k, x, r, dot_r, dot_zero, alpha, omega, beta, dotr) -- traces have length k."""
    row, col, val = _coo(row, col, val)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    r = np.array(b, dtype=np.float64)
    tr = [np.zeros(max(max_iter, 1)) for _ in range(4)]
    o = OrcOpts(tol, max_iter, krr, nrr, _d(tr[0]), _d(tr[1]), _d(tr[2]), _d(tr[3]), 0.0, 0.0)
    if counts is not None:
        keep, cp = _counts(counts, nranks)
        k = lib().orc_solve_coo_part(METHODS[method], nranks, cp, n, len(val), _u(row), _u(col), _d(val), _d(x),
                                     _d(r), C.byref(o))
    else:
        k = lib().orc_solve_coo(METHODS[method], nranks, n, len(val), _u(row), _u(col), _d(val), _d(x),
                                _d(r), C.byref(o))
    return dict(k=k, x=x, r=r, dot_r=o.dot_r, dot_zero=o.dot_zero, alpha=tr[0][:k], omega=tr[1][:k],
                beta=tr[2][:k], dotr=tr[3][:k])


SHIFTED = {"shifted_lopbicgstab": 0, "shifted_pipe_lopbicgstab": 1, "shifted_bicgstab": 2}


def solve_shifted(n, row, col, val, b, sigma, seed, nranks=1, tol=1e-12, max_iter=1000, which="shifted_lopbicgstab"):
    """shifted oracle: returns dict(k, x [nsig][n], r, dot_r, dot_zero, alpha, omega, beta, dotr) (seed scalars)."""
    row, col, val = _coo(row, col, val)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    x = np.zeros(len(sigma) * n)
    r = np.array(b, dtype=np.float64)
    tr = [np.zeros(max(max_iter, 1)) for _ in range(4)]
    o = OrcOpts(tol, max_iter, 0, 0, _d(tr[0]), _d(tr[1]), _d(tr[2]), _d(tr[3]), 0.0, 0.0)
    k = lib().orc_shifted_coo(SHIFTED[which], nranks, n, len(val), _u(row), _u(col), _d(val), _d(x), _d(r), _d(sigma),
                              len(sigma), seed, C.byref(o))
    return dict(k=k, x=x.reshape(len(sigma), n), r=r, dot_r=o.dot_r, dot_zero=o.dot_zero, alpha=tr[0][:k],
                omega=tr[1][:k], beta=tr[2][:k], dotr=tr[3][:k])


def solve_switching(n, row, col, val, b, sigma, seed, nranks=1, tol=1e-12, max_iter=1000, which="shifted_lopbicg_switching"):
    """Oracle restatement of reference src/shifted_switching_solver.c: shifted_lopbicg (per-shift stop
    flags) or shifted_lopbicg_switching / _noovlp (seed switching). k as the reference returns it."""
    row, col, val = _coo(row, col, val)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    nsig = len(sigma)
    x = np.zeros(nsig * n)
    r = np.array(b, dtype=np.float64)
    tr = [np.zeros(max(max_iter, 1)) for _ in range(4)]
    o = OrcOpts(tol, max_iter, 0, 0, _d(tr[0]), _d(tr[1]), _d(tr[2]), _d(tr[3]), 0.0, 0.0)
    info = np.zeros(2 + nsig, dtype=np.int32)
    fn = lib().orc_switching_coo
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_uint, _up, _up, _dp, _dp, _dp, _dp, C.c_int, C.c_int,
                   C.POINTER(OrcOpts), C.POINTER(C.c_int)]
    k = fn(0 if which == "shifted_lopbicg" else 1, nranks, n, len(val), _u(row), _u(col), _d(val), _d(x), _d(r), _d(sigma),
           nsig, seed, C.byref(o), info.ctypes.data_as(C.POINTER(C.c_int)))
    its = k if which == "shifted_lopbicg" else k - 1
    return dict(k=k, x=x.reshape(nsig, n), r=r, dot_r=o.dot_r, dot_zero=o.dot_zero, final_seed=int(info[0]),
                switches=int(info[1]), stop=info[2:].astype(bool), alpha=tr[0][:its], omega=tr[1][:its], beta=tr[2][:its],
                dotr=tr[3][:its])
