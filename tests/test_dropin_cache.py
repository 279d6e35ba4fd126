"""`-m gpu`: matrix residency across drop-in calls (SURVEY.md section 8b "Ownership"; the reference's drivers call
a solver 10-28 x on one matrix, src/main_repeat.c:109-132). The second and later calls with unchanged blocks skip
plan + upload; results are bit-identical; an in-place edit of the matrix (what csr_shift_diagonal does,
src/matrix.c:518-531) is noticed."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from mpi_bicgstab_amd import hipsolver as H
from mpi_bicgstab_amd import synth

pytestmark = pytest.mark.gpu
_dp = C.POINTER(C.c_double)


def _stats():
    h, m = C.c_uint(0), C.c_uint(0)
    H.lib().bicg_dropin_stats(C.byref(h), C.byref(m))
    return h.value, m.value


def test_repeated_dropin_calls_reuse_the_resident_matrix():
    L = H.lib()
    L.bicg_comm_init_single(0)
    os.environ["BICG_MAX_ITER"] = "6"
    os.environ["BICG_QUIET"] = "1"
    try:
        A = synth.transport_like(n=200264, scale_decades=2.0)        # one 8-GPU rank's worth of Transport
        blk = H.single_rank_blocks(A)
        b = A.matvec(np.ones(A.rows))
        L.bicg_dropin_release()
        h0, m0 = _stats()

        def call(fn=L.bicgstab):
            x, r = np.zeros(A.rows), b.copy()
            t0 = time.perf_counter()
            k = fn(C.byref(blk.diag), C.byref(blk.offd), C.byref(blk.info), x.ctypes.data_as(_dp), r.ctypes.data_as(_dp))
            return time.perf_counter() - t0, k, x, r

        t1, k1, x1, r1 = call()
        times = [call() for _ in range(9)]                           # main_repeat.c: 10 solves of one system
        assert _stats() == (h0 + 9, m0 + 1), "one upload for ten calls"
        for t, k, x, r in times:
            assert k == k1 and np.array_equal(x, x1) and np.array_equal(r, r1)
        t2 = min(t for t, *_ in times)
        assert t2 < 0.5 * t1, (t1, t2)      # small matrix: the solve itself is most of a repeated call (full size: test_full_size.py)
        # another solver on the same blocks: still resident
        L.pipe_bicgstab.argtypes = L.bicgstab.argtypes
        call(L.pipe_bicgstab)
        assert _stats() == (h0 + 10, m0 + 1)
        # in-place edit of one value (same arrays, same sizes): must be re-uploaded, and the answer changes
        keep = blk._keep[0]
        assert keep.ctypes.data == C.cast(blk.diag.val, C.c_void_p).value
        keep[12345] *= 1.5
        t3, k3, x3, r3 = call()
        assert _stats() == (h0 + 10, m0 + 2)
        assert not np.array_equal(x3, x1)
        # the handle the host uses for b = A*1 is the same resident context
        ctx = L.bicg_dropin_context(C.byref(blk.diag), C.byref(blk.offd), C.byref(blk.info))
        assert ctx and _stats() == (h0 + 11, m0 + 2)
    finally:
        os.environ.pop("BICG_MAX_ITER", None)
        os.environ.pop("BICG_QUIET", None)
        L.bicg_dropin_release()
