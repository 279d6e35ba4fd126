"""ctypes binding of libbicgstab_hip.so (include/bicgstab_hip.h) -- the host-side mirror used by
tests/ and bench.py. The reference's host is C (src/main.c); the equivalent C host lives in
``mpi-bicgstab_amd/host/``. Function names follow the reference's solver.h.

There is NO fallback: if the HIP library is missing or no GPU is visible the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool's host driver (peer-to-peer mailboxes, RCCL)

from .synth import CSR

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("BICG_HIP_LIB") or os.path.join(PKG_DIR, "libbicgstab_hip.so")

METHODS = {"bicgstab": 0, "ca_bicgstab": 1, "pipe_bicgstab": 2, "pipe_bicgstab_rr": 3}

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint)
_ip = C.POINTER(C.c_int)


class CSRMatrix(C.Structure):      # include/bicgstab_hip.h (reference src/matrix.h:19-26)
    _fields_ = [("val", _dp), ("col", _up), ("ptr", _up), ("nz", C.c_uint), ("rows", C.c_uint), ("cols", C.c_uint)]


class InfoMatrix(C.Structure):     # include/bicgstab_hip.h (reference src/matrix.h:28-33)
    _fields_ = [("nz", C.c_uint), ("rows", C.c_uint), ("cols", C.c_uint), ("code", C.c_char * 4),
                ("recvcounts", _ip), ("displs", _ip)]


class Options(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int), ("out_iter", C.c_int), ("check_every", C.c_int),
                ("quiet", C.c_int), ("krr", C.c_int), ("nrr", C.c_int), ("record_trace", C.c_int),
                ("time_kernels", C.c_int), ("rr_drift", C.c_double)]


class Result(C.Structure):
    _fields_ = [("iterations", C.c_int), ("dot_r", C.c_double), ("dot_zero", C.c_double), ("seconds", C.c_double),
                ("iter_seconds", C.c_double), ("spmv_ms_total", C.c_double), ("spmv_launches", C.c_int),
                ("breakdown_iteration", C.c_int), ("adaptive_replacements", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(None, _dp, C.c_int, C.c_void_p)
ALLTOALLV_FN = C.CFUNCTYPE(None, C.c_void_p, _ip, _ip, C.c_void_p, _ip, _ip, C.c_void_p)

EXPORTS = [
    "bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr",
    "shifted_bicgstab", "shifted_lopbicgstab", "shifted_lopbicgstab_v2", "shifted_lopbicgstab_nooverlap",
    "shifted_pipe_lopbicgstab", "shifted_pipe_lopbicgstab_nooverlap", "bicg_solve_shifted",
    "shifted_lopbicg", "shifted_lopbicg_switching", "shifted_lopbicg_switching_noovlp", "bicg_shifted_residuals",
    "bicg_comm_enable_p2p", "bicg_comm_p2p_active", "bicg_comm_failed",
    "bicg_partition_nnz", "bicg_mtx_load_block_part", "bicg_mtx_parse_double", "bicg_mtx_cache_save", "bicg_mtx_cache_load",
    "bicg_coo_to_blocks_device", "bicg_mtx_set_block_builder",
    "bicg_comm_unique_id", "bicg_comm_init_rccl", "bicg_comm_init_host", "bicg_comm_init_mpi",
    "bicg_comm_init_single", "bicg_comm_finalize", "bicg_comm_selftest_rccl", "bicg_comm_rccl_loadable", "bicg_comm_last_error", "bicg_section_times", "bicg_comm_rank", "bicg_comm_size",
    "bicg_default_options", "bicg_create", "bicg_destroy", "bicg_solve", "bicg_load", "bicg_run", "bicg_fetch",
    "bicg_run_begin", "bicg_run_iterate", "bicg_run_iterate_timed", "bicg_run_end", "bicg_sync", "bicg_trace", "bicg_spmv", "bicg_dot", "bicg_spmv_bench", "bicg_plan_info", "bicg_ctx_flags", "bicg_spmm", "bicg_device_matrix_bytes", "bicg_uniform_entries", "bicg_constant_entries", "bicg_masked_rows", "bicg_stencil_info", "bicg_stencil_rows_per_lane", "bicg_comm_wait_stats", "bicg_plan_collisions", "bicg_product_kernels", "bicg_spmv_matrix_bytes", "bicg_last_shifted_persistent", "bicg_last_spmm_windowed", "bicg_dropin_context", "bicg_dropin_release", "bicg_dropin_stats",
    "bicg_mtx_load_block", "bicg_mtx_free", "bicg_partition", "bicg_halo_plan", "bicg_halo_send_counts", "bicg_halo_send_lists", "bicg_row_blocks", "bicg_window_plan", "bicg_window_slot", "bicg_version", "bicg_has_experiments", "bicg_switch_value", "bicg_switch_unknown", "bicg_stream_bench", "bicg_create_device_csr", "bicg_stencil7_device", "bicg_device_free", "bicg_persist_plan", "bicg_set_plan_threads",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C mpi-bicgstab_amd lib` "
                               "(python -c 'import __graft_entry__ as g; g.build()')")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L.bicg_create.restype = C.c_void_p
        L.bicg_create.argtypes = [C.POINTER(CSRMatrix), C.POINTER(CSRMatrix), C.POINTER(InfoMatrix)]
        L.bicg_destroy.argtypes = [C.c_void_p]
        L.bicg_solve.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.POINTER(Options), C.POINTER(Result)]
        L.bicg_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(Options), C.POINTER(Result)]
        L.bicg_solve_shifted.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.POINTER(Options),
                                         C.POINTER(Result)]
        L.bicg_run_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(Options)]
        L.bicg_run_iterate.argtypes = [C.c_void_p, C.c_int]
        L.bicg_run_iterate_timed.argtypes = [C.c_void_p, C.c_int, _dp]
        L.bicg_switch_value.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.bicg_switch_unknown.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.bicg_run_end.argtypes = [C.c_void_p, C.POINTER(Result)]
        L.bicg_sync.argtypes = [C.c_void_p]
        L.bicg_load.argtypes = [C.c_void_p, _dp, _dp]
        L.bicg_fetch.argtypes = [C.c_void_p, _dp, _dp]
        L.bicg_trace.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.bicg_spmv.argtypes = [C.c_void_p, _dp, _dp]
        L.bicg_dot.argtypes = [C.c_void_p, _dp, _dp]
        L.bicg_dot.restype = C.c_double
        L.bicg_spmv_bench.argtypes = [C.c_void_p, C.c_int, _dp]
        L.bicg_plan_info.argtypes = [C.c_void_p, _up]
        L.bicg_stencil_info.argtypes = [C.c_void_p, _up]
        L.bicg_stencil_rows_per_lane.argtypes = [C.c_void_p]; L.bicg_stencil_rows_per_lane.restype = C.c_uint
        L.bicg_comm_wait_stats.argtypes = [C.c_void_p, _dp]
        L.bicg_section_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.bicg_comm_failed.argtypes = [C.c_void_p]
        L.bicg_dropin_context.restype = C.c_void_p
        L.bicg_dropin_context.argtypes = [C.POINTER(CSRMatrix), C.POINTER(CSRMatrix), C.POINTER(InfoMatrix)]
        L.bicg_dropin_stats.argtypes = [_up, _up]
        L.bicg_spmm.argtypes = [C.c_void_p, _dp, _dp, C.c_int, _dp, _dp]
        L.bicg_ctx_flags.argtypes = [C.c_void_p]
        L.bicg_device_matrix_bytes.argtypes = [C.c_void_p]
        L.bicg_device_matrix_bytes.restype = C.c_ulonglong
        L.bicg_ctx_flags.restype = C.c_uint
        for fn in (L.bicg_uniform_entries, L.bicg_constant_entries, L.bicg_masked_rows, L.bicg_spmv_matrix_bytes):
            fn.argtypes = [C.c_void_p]; fn.restype = C.c_ulonglong
        L.bicg_plan_collisions.argtypes = [C.c_void_p]; L.bicg_plan_collisions.restype = C.c_uint
        L.bicg_product_kernels.argtypes = [C.c_int]; L.bicg_product_kernels.restype = C.c_uint
        L.bicg_last_shifted_persistent.argtypes = [C.c_void_p]
        L.bicg_last_spmm_windowed.argtypes = [C.c_void_p]
        L.bicg_shifted_residuals.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int, _dp]
        L.bicg_default_options.argtypes = [C.POINTER(Options)]
        L.bicg_comm_unique_id.argtypes = [C.c_void_p]
        L.bicg_comm_init_rccl.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.bicg_comm_init_host.argtypes = [C.c_int, C.c_int, ALLREDUCE_FN, ALLTOALLV_FN, C.c_void_p, C.c_int]
        L.bicg_comm_init_single.argtypes = [C.c_int]
        L.bicg_comm_init_mpi.argtypes = [C.c_char_p, C.c_int]
        L.bicg_partition.argtypes = [C.c_uint, C.c_int, _ip, _ip]
        L.bicg_halo_plan.argtypes = [C.POINTER(CSRMatrix), C.POINTER(InfoMatrix), C.c_int, C.c_uint, _up, _ip, _up]
        L.bicg_halo_send_counts.argtypes = [C.c_int, _ip, ALLTOALLV_FN, C.c_void_p, _ip]
        L.bicg_halo_send_lists.argtypes = [C.c_int, C.c_int, C.POINTER(InfoMatrix), C.c_uint, _up, _ip, _ip, ALLTOALLV_FN,
                                           C.c_void_p, _up]
        L.bicg_row_blocks.argtypes = [_up, C.c_uint, C.c_uint, C.c_uint, _up]
        L.bicg_row_blocks.restype = C.c_uint
        L.bicg_window_plan.argtypes = [_up, _up, C.c_uint, C.c_uint, C.c_char_p, C.c_uint, C.c_uint, _up, _up, _up]
        L.bicg_window_plan.restype = C.c_long
        L.bicg_window_slot.argtypes = [_up, C.c_uint, C.c_uint, C.c_uint]
        L.bicg_window_slot.restype = C.c_uint
        L.bicg_version.restype = C.c_char_p
        L.bicg_stream_bench.argtypes = [C.c_int, C.c_ulonglong, C.c_int, _dp, _dp]
        L.bicg_create_device_csr.restype = C.c_void_p
        L.bicg_create_device_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, _dp]
        L.bicg_stencil7_device.argtypes = [C.c_uint, _dp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_ulonglong)]
        L.bicg_device_free.argtypes = [C.c_void_p]
        for name in ("bicgstab", "ca_bicgstab", "pipe_bicgstab"):
            getattr(L, name).argtypes = [C.POINTER(CSRMatrix), C.POINTER(CSRMatrix), C.POINTER(InfoMatrix), _dp, _dp]
        L.pipe_bicgstab_rr.argtypes = [C.POINTER(CSRMatrix), C.POINTER(CSRMatrix), C.POINTER(InfoMatrix), _dp, _dp,
                                       C.c_int, C.c_int]
        _lib = L
    return _lib


# The library's token-list variables (csrc/bicg_knobs.h): keyword -> (variable, token). INTEGRATION.md section 6 says what each does.
SWITCHES = {k: ("BICG_PLAN", k.replace("_", "-")) for k in (
    "stencil", "lines", "planes", "ca_fuse", "layout", "window", "col16", "uniform", "constant", "masked", "desc", "lists", "jagw",
    "spmm", "spmm_window", "fuse_pipe", "pipe_probe", "halo_fused", "window_list", "wide")}
SWITCHES.update(persist=("BICG_PERSIST", "0"), persist_chunk=("BICG_PERSIST", "chunk"), persist_shifted=("BICG_PERSIST", "shifted"),
                force_comm=("BICG_TEST", "force-comm"), spin_ticks=("BICG_TEST", "spin-ticks"),
                p2p_fault_after=("BICG_TEST", "p2p-fault-after"), plan_collide=("BICG_TEST", "plan-collide"))
SWITCH_VARS = ("BICG_PLAN", "BICG_PERSIST", "BICG_TEST")


def switches(env=None, **kw):
    """Set (value) or clear (None) tokens of BICG_PLAN / BICG_PERSIST / BICG_TEST in `env` (default: this process's environment),
    the other tokens stay: switches(stencil=0, lines=2) -> BICG_PLAN="stencil=0,lines=2"; switches(force_comm=1) ->
    BICG_TEST="force-comm=1"; switches(persist=0) -> BICG_PERSIST="0" beside its chunk= / shifted= tokens, persist=1 or None
    takes the "0" away. Contexts read the variables when they are created."""
    env = os.environ if env is None else env
    for k, v in kw.items():
        var, tok = SWITCHES[k]
        toks = [t for t in env.get(var, "").split(",") if t and t.split("=")[0] != tok and not (k == "persist" and t in ("1", "off"))]
        if k == "persist":
            if v is not None and int(v) == 0:
                toks.insert(0, "0")
        elif v is not None:
            toks.append(f"{tok}={v}")
        if toks:
            env[var] = ",".join(toks)
        else:
            env.pop(var, None)
    return env


def switch_value(var: str, name: str):
    """what the LIBRARY reads for token `name` of BICG_PLAN / BICG_PERSIST / BICG_TEST (bicg_switch_value): its text, None if absent"""
    buf = C.create_string_buffer(64)
    n = lib().bicg_switch_value(var.encode(), name.encode(), buf, 64)
    return None if n < 0 else buf.value.decode()


def _d(a):
    return a.ctypes.data_as(_dp)


def window_plan(A: CSR, group_rows: int = 256, max_slots: int = 4096, gap: int = 8):
    """x windows of a block (bicg_window_plan): (win_ptr, runs[n][2], slots of the largest window) or None when a
    group does not fit max_slots"""
    ptr = np.ascontiguousarray(A.ptr, dtype=np.uint32)
    col = np.ascontiguousarray(A.col, dtype=np.uint32)
    up = lambda a: a.ctypes.data_as(_up)
    n = lib().bicg_window_plan(up(ptr), up(col), A.rows, group_rows, None, max_slots, gap, None, None, None)
    if n < 0:
        return None
    ngroups = (A.rows + group_rows - 1) // group_rows
    win_ptr = np.zeros(ngroups + 1, dtype=np.uint32)
    runs = np.zeros((max(n, 1), 2), dtype=np.uint32)
    used = C.c_uint(0)
    lib().bicg_window_plan(up(ptr), up(col), A.rows, group_rows, None, max_slots, gap, up(win_ptr), up(runs), C.byref(used))
    return win_ptr, runs[:n], int(used.value)


class HostBlocks:
    """One rank's diag/offd blocks + INFO in the reference's struct layout (keeps numpy buffers alive)."""

    def __init__(self, diag: CSR, offd: CSR | None, n_global: int, counts, displs):
        if offd is None:
            offd = CSR(diag.rows, n_global, np.zeros(diag.rows + 1, dtype=np.uint32), np.zeros(1, dtype=np.uint32),
                       np.zeros(1))
        self._keep = []

        def mk(A: CSR):
            v = np.ascontiguousarray(A.val, dtype=np.float64)
            c = np.ascontiguousarray(A.col, dtype=np.uint32)
            p = np.ascontiguousarray(A.ptr, dtype=np.uint32)
            if v.size == 0:
                v = np.zeros(1)
            if c.size == 0:
                c = np.zeros(1, dtype=np.uint32)
            self._keep += [v, c, p]
            return CSRMatrix(_d(v), c.ctypes.data_as(_up), p.ctypes.data_as(_up), int(p[-1]), A.rows, A.cols)

        self.diag, self.offd = mk(diag), mk(offd)
        cnt = np.ascontiguousarray(counts, dtype=np.int32)
        dsp = np.ascontiguousarray(displs, dtype=np.int32)
        self._keep += [cnt, dsp]
        self.info = InfoMatrix(0, n_global, n_global, b"MCRG", cnt.ctypes.data_as(_ip), dsp.ctypes.data_as(_ip))
        self.n_loc = diag.rows


def load_mtx_blocks(path: str, rank: int = 0, nranks: int = 1) -> HostBlocks:
    """This rank's blocks of a Matrix-Market file through the library's loader (C, reads the file once)."""
    d, o, info = CSRMatrix(), CSRMatrix(), InfoMatrix()
    L = lib()
    L.bicg_mtx_load_block.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(CSRMatrix), C.POINTER(CSRMatrix), C.POINTER(InfoMatrix)]
    if L.bicg_mtx_load_block(path.encode(), rank, nranks, C.byref(d), C.byref(o), C.byref(info)) != 0:
        raise RuntimeError(f"cannot load {path}")

    def to_csr(m, ncols):
        nz = int(m.ptr[m.rows])
        return CSR(m.rows, ncols, np.ctypeslib.as_array(m.ptr, shape=(m.rows + 1,)).copy(),
                   np.ctypeslib.as_array(m.col, shape=(max(nz, 1),))[:nz].copy(),
                   np.ctypeslib.as_array(m.val, shape=(max(nz, 1),))[:nz].copy())
    diag, offd = to_csr(d, d.rows), to_csr(o, info.cols)
    counts = np.ctypeslib.as_array(info.recvcounts, shape=(nranks,)).copy()
    displs = np.ctypeslib.as_array(info.displs, shape=(nranks,)).copy()
    n = int(info.rows)
    L.bicg_mtx_free(C.byref(d), C.byref(o), C.byref(info))
    blk = HostBlocks(diag, offd if nranks > 1 else None, n, counts, displs)
    blk.nnz_global = int(info.nz)
    return blk


def single_rank_blocks(A: CSR) -> HostBlocks:
    return HostBlocks(A, None, A.rows, [A.rows], [0])


class Context:
    """bicg_ctx wrapper: matrix resident on the GPU; solves, SpMV, dot."""

    def __init__(self, blocks: HostBlocks):
        self.blocks = blocks
        self.n = blocks.n_loc
        self.h = lib().bicg_create(C.byref(blocks.diag), C.byref(blocks.offd), C.byref(blocks.info))
        if not self.h:
            raise RuntimeError("bicg_create failed")

    @classmethod
    def stencil7_on_device(cls, m: int, weights):
        """Single rank: the 7-point stencil on an m^3 grid generated, planned and kept on the GPU (bicg_stencil7_device +
        bicg_create_device_csr) -- no host copy of the matrix. Returns (context, nnz, plan seconds, generation seconds)."""
        import time
        L = lib()
        w = np.ascontiguousarray(weights, dtype=np.float64)
        val, col, ptr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nnz = C.c_ulonglong(0)
        t0 = time.perf_counter()
        if L.bicg_stencil7_device(m, _d(w), C.byref(val), C.byref(col), C.byref(ptr), C.byref(nnz)) != 0:
            raise RuntimeError("bicg_stencil7_device failed")
        t_gen = time.perf_counter() - t0
        secs = C.c_double(0.0)
        self = cls.__new__(cls)
        self.blocks = None
        self.n = m ** 3
        self.h = L.bicg_create_device_csr(val, col, ptr, m ** 3, C.byref(secs))
        for p in (val, col, ptr):
            L.bicg_device_free(p)
        if not self.h:
            raise RuntimeError("bicg_create_device_csr refused the matrix")
        return self, int(nnz.value), float(secs.value), t_gen

    def close(self):
        if self.h:
            lib().bicg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def options(self, **kw) -> Options:
        o = Options()
        lib().bicg_default_options(C.byref(o))
        for k, val in kw.items():
            setattr(o, k, val)
        return o

    def solve(self, method: str, b, x0=None, **kw):
        """Semantics of the reference solver call: returns dict(k, x, r, result)."""
        x = np.zeros(self.n) if x0 is None else np.array(x0, dtype=np.float64)
        r = np.array(b, dtype=np.float64)
        kw.setdefault("quiet", 1)
        o = self.options(**kw)
        res = Result()
        k = lib().bicg_solve(self.h, METHODS[method], _d(x), _d(r), C.byref(o), C.byref(res))
        return dict(k=k, x=x, r=r, dot_r=res.dot_r, dot_zero=res.dot_zero, result=res)

    SHIFTED = {"shifted_lopbicgstab": 0, "shifted_pipe_lopbicgstab": 1, "shifted_bicgstab": 2,
               "shifted_lopbicg": 3, "shifted_lopbicg_switching": 4}

    def solve_shifted(self, b, sigma, seed, x0_set=None, which="shifted_lopbicgstab", **kw):
        """(A + sigma_j I) x_j = b for all j (reference src/shifted_solver.c): dict(k, x [nsig][n], r, result)."""
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        x = np.zeros((len(sigma), self.n)) if x0_set is None else np.array(x0_set, dtype=np.float64).reshape(len(sigma), self.n)
        r = np.array(b, dtype=np.float64)
        kw.setdefault("quiet", 1)
        kw.setdefault("tol", 1e-12)
        o = self.options(**kw)
        res = Result()
        k = lib().bicg_solve_shifted(self.h, self.SHIFTED[which], _d(x), _d(r), _d(sigma), len(sigma), seed, C.byref(o),
                                     C.byref(res))
        return dict(k=k, x=x, r=r, dot_r=res.dot_r, dot_zero=res.dot_zero, result=res,
                    iterations=res.iterations, switches=res.adaptive_replacements)

    def section_times(self):
        """Section times of the last solve run with time_kernels=2 (the reference's MEASURE_SECTION_TIME, on the device
        clock): dict(vec_ms, spmv_ms, shift_ms, reduce_ms, iterations, marks) or None when that solve was not timed."""
        ms = (C.c_double * 4)()
        it, marks = C.c_int(0), C.c_int(0)
        if lib().bicg_section_times(self.h, ms, C.byref(it), C.byref(marks)) != 0:
            return None
        return dict(vec_ms=ms[0], spmv_ms=ms[1], shift_ms=ms[2], reduce_ms=ms[3], iterations=it.value, marks=marks.value)

    def shifted_residuals(self, x_set, b, sigma):
        """|| (A + sigma_j I) x_j - b || / || b || for every shift (reference src/test_shifted.c:129-154)"""
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        x = np.ascontiguousarray(x_set, dtype=np.float64).reshape(len(sigma), self.n)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros(len(sigma))
        lib().bicg_shifted_residuals(self.h, _d(x), _d(b), _d(sigma), len(sigma), _d(out))
        return out

    def spmm(self, x_set, sigma=None):
        """Y_j = (A + sigma_j I) X_j for all rows of x_set [nvec][n], A read once per 16 vectors -> (Y, device ms)"""
        x = np.ascontiguousarray(x_set, dtype=np.float64).reshape(-1, self.n)
        y = np.zeros_like(x)
        sg = None if sigma is None else np.ascontiguousarray(sigma, dtype=np.float64)
        ms = C.c_double(0.0)
        rc = lib().bicg_spmm(self.h, _d(x), None if sg is None else _d(sg), x.shape[0], _d(y), C.byref(ms))
        if rc != 0:
            raise RuntimeError("bicg_spmm: the matrix is not entirely on the sliced-ELL path")
        return y, ms.value

    def load(self, x0, b):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        lib().bicg_load(self.h, _d(x0), _d(b))

    def run(self, method: str, **kw) -> Result:
        kw.setdefault("quiet", 1)
        o = self.options(**kw)
        res = Result()
        lib().bicg_run(self.h, METHODS[method], C.byref(o), C.byref(res))
        return res

    def run_begin(self, method: str, **kw):
        kw.setdefault("quiet", 1)
        self._opt = self.options(**kw)
        lib().bicg_run_begin(self.h, METHODS[method], C.byref(self._opt))

    def run_iterate(self, nsteps: int) -> int:
        return lib().bicg_run_iterate(self.h, nsteps)

    def run_iterate_timed(self, nsteps: int):
        """run_iterate with three clocks (ms): device events around the launches, host time spent enqueueing, host wall time"""
        ms = (C.c_double * 3)()
        k = lib().bicg_run_iterate_timed(self.h, nsteps, ms)
        return k, dict(device_ms=ms[0], enqueue_ms=ms[1], wall_ms=ms[2])

    def run_end(self) -> Result:
        res = Result()
        lib().bicg_run_end(self.h, C.byref(res))
        return res

    def sync(self):
        lib().bicg_sync(self.h)

    def fetch(self):
        x, r = np.zeros(self.n), np.zeros(self.n)
        lib().bicg_fetch(self.h, _d(x), _d(r))
        return x, r

    def trace(self, k: int):
        arrs = [np.zeros(max(k, 1)) for _ in range(4)]
        got = lib().bicg_trace(self.h, *[_d(a) for a in arrs])
        return dict(zip(("alpha", "omega", "beta", "dotr"), [a[:got] for a in arrs]))

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.n)
        lib().bicg_spmv(self.h, _d(x), _d(y))
        return y

    def dot(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        return lib().bicg_dot(self.h, _d(x), _d(y))

    def spmv_bench(self, reps: int) -> float:
        ms = C.c_double(0.0)
        lib().bicg_spmv_bench(self.h, reps, C.byref(ms))
        return ms.value

    def comm_failed(self) -> bool:
        return bool(lib().bicg_comm_failed(self.h))

    FLAGS = {"p2p": 1, "ll_fused": 2, "overlap": 4, "col16": 8, "all_sell": 16, "jagged": 32, "spmm": 64, "window": 128, "rowsplit": 256, "persist": 512,
             "fuse_pipe": 1024, "pipe_probed": 2048, "uniform": 4096, "constant": 8192}

    def flags(self):
        f = int(lib().bicg_ctx_flags(self.h))
        return {k: bool(f & v) for k, v in self.FLAGS.items()}

    def device_matrix_bytes(self) -> int:
        return int(lib().bicg_device_matrix_bytes(self.h))

    def uniform_entries(self) -> int:
        return int(lib().bicg_uniform_entries(self.h))

    def constant_entries(self) -> int:
        return int(lib().bicg_constant_entries(self.h))

    def masked_rows(self) -> int:
        return int(lib().bicg_masked_rows(self.h))

    def comm_wait_stats(self):
        """microseconds the exchanges of the last solve made the persistent kernels wait (None: nothing recorded)"""
        out = (C.c_double * 6)()
        if lib().bicg_comm_wait_stats(self.h, out) != 0:
            return None
        return dict(zip(("allreduce_p50_us", "allreduce_p99_us", "handoff_p50_us", "handoff_p99_us", "allreduce_samples", "handoff_samples"), list(out)))

    def stencil_info(self):
        """The plane-marching product of a 7-point grid stencil (csrc/bicg_stencil.hip): is it in use, and its tiling."""
        out = (C.c_uint * 8)()
        lib().bicg_stencil_info(self.h, out)
        info = dict(zip(("on", "sy", "ny", "nz", "lines", "planes", "workgroups", "masked_segments"), list(out)))
        info["rows_per_lane"] = int(lib().bicg_stencil_rows_per_lane(self.h))
        return info

    def plan_collisions(self) -> int:
        return int(lib().bicg_plan_collisions(self.h))

    def last_spmm_windowed(self) -> bool:
        return bool(lib().bicg_last_spmm_windowed(self.h))

    def last_spmm_kind(self) -> str:
        """which SpMM kernel the last pass ran: "pipelined" (k_spmm_pipe), "windowed" (k_spmm_win) or "rowmajor" (k_spmm_sell)"""
        return ("rowmajor", "windowed", "pipelined")[int(lib().bicg_last_spmm_windowed(self.h))]

    def last_shifted_persistent(self) -> bool:
        return bool(lib().bicg_last_shifted_persistent(self.h))

    def spmv_matrix_bytes(self) -> int:
        return int(lib().bicg_spmv_matrix_bytes(self.h))

    def plan_info(self):
        out = (C.c_uint * 8)()
        lib().bicg_plan_info(self.h, out)
        return dict(zip(("rows", "nnz_diag", "nnz_offd", "halo", "row_blocks", "boundary_blocks", "sell_rows",
                         "sell_padding"), list(out)))


PRODUCT_KERNELS = {"sell_padded": 1, "sell_jagged": 2, "sell_window_loop": 4, "jagw": 8, "stencil": 16, "csr": 32, "rows": 64, "sell_epilogue": 128,
                   "jagd": 512, "jagw_list": 1024}


def product_kernels(reset: bool = True):
    """names of the product kernels launched since the last reset (bicg_product_kernels)"""
    m = int(lib().bicg_product_kernels(1 if reset else 0))
    return sorted(k for k, v in PRODUCT_KERNELS.items() if m & v)


STREAM_KINDS = {"copy": 0, "triad": 1, "read8": 2, "read16": 3}


def stream_bench(kind: str, bytes_per_array: int = 1 << 30, reps: int = 20) -> float:
    """GB/s of a STREAM-style pass on the current GPU (bicg_stream_bench): copy / triad / read8 / read16"""
    g, ms = C.c_double(0.0), C.c_double(0.0)
    if lib().bicg_stream_bench(STREAM_KINDS[kind], bytes_per_array, reps, C.byref(g), C.byref(ms)) != 0:
        raise RuntimeError("bicg_stream_bench failed")
    return g.value


# ---- host-only helpers (no GPU) ----------------------------------------------------------------
def persist_plan(blocks: HostBlocks, nranks: int, gmax: int):
    """Plan of the persistent iteration (bicg_persist_plan) for one rank's blocks -> dict, or None when the block does not
    qualify. With nranks > 1 the offd block is renumbered to [rows + halo position] through bicg_halo_plan first."""
    L = lib()
    _usp = C.POINTER(C.c_ushort)
    L.bicg_persist_plan.argtypes = [C.POINTER(CSRMatrix), C.POINTER(CSRMatrix), C.c_uint, _up, _up, _usp, _dp, _usp, _usp, _up, _up]
    offd_p, keep = None, None
    halo = 0
    if nranks > 1:
        halo, halo_cols, rc, ren = halo_plan(blocks, nranks)
        nz = int(blocks.offd.ptr[blocks.offd.rows])
        ren = np.ascontiguousarray(ren[:max(nz, 1)], dtype=np.uint32)
        keep = (ren,)
        o = CSRMatrix(blocks.offd.val, ren.ctypes.data_as(_up), blocks.offd.ptr, nz, blocks.offd.rows, blocks.n_loc + halo)
        offd_p = C.byref(o)
    summ = (C.c_uint * 8)()
    if L.bicg_persist_plan(C.byref(blocks.diag), offd_p, gmax, summ, None, None, None, None, None, None, None) != 0:
        return None
    spw, nwg, win_slots, max_runs, max_entries, entries, nruns, rpt = list(summ)      # spw: slices per workgroup; rpt: rows per thread
    n = blocks.n_loc
    nslices = (n + 63) // 64
    pbase = np.zeros(nslices + 1, dtype=np.uint32)
    pslot = np.zeros(max(entries, 1), dtype=np.uint16)
    pval = np.zeros(max(entries, 1))
    rlen = np.zeros(n, dtype=np.uint16)
    rdiag = np.zeros(n, dtype=np.uint16)
    wptr = np.zeros(nwg + 1, dtype=np.uint32)
    runs = np.zeros(2 * max(nruns, 1), dtype=np.uint32)
    rc2 = L.bicg_persist_plan(C.byref(blocks.diag), offd_p, gmax, summ, pbase.ctypes.data_as(_up), pslot.ctypes.data_as(_usp), _d(pval),
                              rlen.ctypes.data_as(_usp), rdiag.ctypes.data_as(_usp), wptr.ctypes.data_as(_up), runs.ctypes.data_as(_up))
    assert rc2 == 0
    return dict(spw=spw, rpt=rpt, nwg=nwg, win_slots=win_slots, max_runs=max_runs, max_entries=max_entries, entries=entries, halo=halo,
                pbase=pbase, pslot=pslot[:entries], pval=pval[:entries], rlen=rlen, rdiag=rdiag, wptr=wptr, runs=runs[:2 * nruns].reshape(-1, 2))


def partition(n: int, nranks: int):
    cnt = np.zeros(nranks, dtype=np.int32)
    dsp = np.zeros(nranks, dtype=np.int32)
    lib().bicg_partition(n, nranks, cnt.ctypes.data_as(_ip), dsp.ctypes.data_as(_ip))
    return cnt, dsp


def halo_plan(blocks: HostBlocks, nranks: int):
    nz = max(int(blocks.offd.nz), 1)
    cols = np.zeros(nz, dtype=np.uint32)
    ren = np.zeros(nz, dtype=np.uint32)
    rc = np.zeros(nranks, dtype=np.int32)
    h = lib().bicg_halo_plan(C.byref(blocks.offd), C.byref(blocks.info), nranks, blocks.n_loc,
                             cols.ctypes.data_as(_up), rc.ctypes.data_as(_ip), ren.ctypes.data_as(_up))
    return h, cols[:h], rc, ren[:int(blocks.offd.nz)]


def row_blocks(ptr, chunk=2048, max_rows=1024):
    ptr = np.ascontiguousarray(ptr, dtype=np.uint32)
    rows = len(ptr) - 1
    out = np.zeros(rows + 1, dtype=np.uint32)
    nb = lib().bicg_row_blocks(ptr.ctypes.data_as(_up), rows, chunk, max_rows, out.ctypes.data_as(_up))
    return out[:nb + 1]
