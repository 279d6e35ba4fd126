// bicg_ingest.hip -- device-side COO -> CSR for one rank's block.
//
// The reference builds a rank's diag/offd CSR blocks on the host: every rank fscanf()s the file
// twice and stable-merge-sorts its triplets by row (src/matrix.c:135-183, 268-396). Here the
// triplets a rank owns (file order, global indices) are uploaded once and turned into the same two
// blocks on the GPU:
//   key = 2 * local_row + (column outside the rank's own range)        one pass
//   STABLE radix sort of (key, position)                                rocPRIM, setup path only
//   per-row counts -> exclusive scans = the two ptr arrays              rocPRIM scan
//   scatter: a sorted entry of row r lands at  i - offd_ptr[r]  (diag)  or  i - diag_ptr[r+1]  (offd)
// A stable sort on the row key keeps the file order inside every row, which is exactly what the
// reference's merge sort produces; the result is bit-identical to the host path
// (tests/test_gpu_parity.py::test_device_ingest_matches_host).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/bicgstab_hip.h"
#include "bicg_comm.h"

namespace {

using namespace bicg;

__global__ void k_keys(const unsigned *row, const unsigned *col, size_t nnz, unsigned lo, unsigned hi, unsigned *key,
                       unsigned *pos, unsigned *cnt_d, unsigned *cnt_o)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < nnz; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned r = row[e] - lo, c = col[e];
        const unsigned off = (c < lo || c >= hi) ? 1u : 0u;
        key[e] = 2u * r + off;
        pos[e] = (unsigned)e;
        atomicAdd(off ? &cnt_o[r] : &cnt_d[r], 1u);
    }
}

__global__ void k_scatter(const unsigned *key_sorted, const unsigned *pos_sorted, size_t nnz, const unsigned *col,
                          const double *val, unsigned lo, const unsigned *dptr, const unsigned *optr, unsigned *dcol,
                          double *dval, unsigned *ocol, double *oval)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned k = key_sorted[i], r = k >> 1, src = pos_sorted[i];
        if (k & 1u) {
            const size_t at = i - dptr[r + 1];       // all diag entries of rows <= r precede it
            ocol[at] = col[src]; oval[at] = val[src];
        } else {
            const size_t at = i - optr[r];           // all offd entries of rows < r precede it
            dcol[at] = col[src] - lo; dval[at] = val[src];
        }
    }
}

template <class T> T *dmalloc(size_t n)
{
    T *p = nullptr;
    BICG_HIP(hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)));
    return p;
}

}  // namespace

extern "C" int bicg_coo_to_blocks_device(const unsigned int *row, const unsigned int *col, const double *val,
                                         unsigned long nnz, unsigned int lo, unsigned int hi, unsigned int ncols,
                                         CSR_Matrix *diag, CSR_Matrix *offd)
{
    using namespace bicg;
    Comm *comm = comm_get();
    BICG_HIP(hipSetDevice(comm->device));
    // rocPRIM reports hipGetLastError() after its launches: an error some EARLIER call of this thread left behind
    // (a probe the caller tolerated, another library's) would be blamed on the sort. Start from a clean slate.
    {
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess && getenv("BICG_DEBUG"))
            fprintf(stderr, "bicgstab_hip: note: clearing an earlier HIP error before the ingest: %s\n", hipGetErrorString(stale));
    }
    const unsigned rows = hi - lo;
    if (nnz >= 0xFFFFFFFFul || rows >= 0x7FFFFFFFu) die("bicg_coo_to_blocks_device", "block too large for 32-bit indices");
    for (unsigned long e = 0; e < nnz; ++e)
        if (row[e] < lo || row[e] >= hi) die("bicg_coo_to_blocks_device", "triplet outside this rank's rows");

    unsigned *d_row = dmalloc<unsigned>(nnz), *d_col = dmalloc<unsigned>(nnz);
    double *d_val = dmalloc<double>(nnz);
    unsigned *key = dmalloc<unsigned>(nnz), *pos = dmalloc<unsigned>(nnz), *key2 = dmalloc<unsigned>(nnz), *pos2 = dmalloc<unsigned>(nnz);
    unsigned *cnt_d = dmalloc<unsigned>((size_t)rows + 1), *cnt_o = dmalloc<unsigned>((size_t)rows + 1);
    unsigned *dptr = dmalloc<unsigned>((size_t)rows + 1), *optr = dmalloc<unsigned>((size_t)rows + 1);
    BICG_HIP(hipMemcpy(d_row, row, sizeof(unsigned) * nnz, hipMemcpyHostToDevice));
    BICG_HIP(hipMemcpy(d_col, col, sizeof(unsigned) * nnz, hipMemcpyHostToDevice));
    BICG_HIP(hipMemcpy(d_val, val, sizeof(double) * nnz, hipMemcpyHostToDevice));
    BICG_HIP(hipMemset(cnt_d, 0, sizeof(unsigned) * ((size_t)rows + 1)));
    BICG_HIP(hipMemset(cnt_o, 0, sizeof(unsigned) * ((size_t)rows + 1)));

    const unsigned grid = (unsigned)((nnz + 255) / 256 < 4096 ? (nnz + 255) / 256 : 4096);
    if (nnz) hipLaunchKernelGGL(k_keys, dim3(grid ? grid : 1), dim3(256), 0, 0, d_row, d_col, (size_t)nnz, lo, hi, key, pos, cnt_d, cnt_o);

    // stable sort by key; only the bits a key can have are sorted
    unsigned bits = 1;
    while (bits < 32 && (1ull << bits) < 2ull * rows) ++bits;
    size_t tmp_bytes = 0, scan_bytes = 0;
    void *tmp = nullptr;
    if (nnz) {
        BICG_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key, key2, pos, pos2, (size_t)nnz, 0u, bits, (hipStream_t)0));
        BICG_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, cnt_d, dptr, 0u, (size_t)rows + 1, rocprim::plus<unsigned>(), (hipStream_t)0));
        if (scan_bytes > tmp_bytes) tmp_bytes = scan_bytes;
        BICG_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
        BICG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, key, key2, pos, pos2, (size_t)nnz, 0u, bits, (hipStream_t)0));
    } else {
        BICG_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, cnt_d, dptr, 0u, (size_t)rows + 1, rocprim::plus<unsigned>(), (hipStream_t)0));
        tmp_bytes = scan_bytes;
        BICG_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
    }
    BICG_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, cnt_d, dptr, 0u, (size_t)rows + 1, rocprim::plus<unsigned>(), (hipStream_t)0));
    BICG_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, cnt_o, optr, 0u, (size_t)rows + 1, rocprim::plus<unsigned>(), (hipStream_t)0));

    unsigned nd = 0, no = 0;
    BICG_HIP(hipMemcpy(&nd, dptr + rows, sizeof(unsigned), hipMemcpyDeviceToHost));
    BICG_HIP(hipMemcpy(&no, optr + rows, sizeof(unsigned), hipMemcpyDeviceToHost));
    unsigned *dcol = dmalloc<unsigned>(nd), *ocol = dmalloc<unsigned>(no);
    double *dval = dmalloc<double>(nd), *oval = dmalloc<double>(no);
    if (nnz) hipLaunchKernelGGL(k_scatter, dim3(grid ? grid : 1), dim3(256), 0, 0, key2, pos2, (size_t)nnz, d_col, d_val, lo, dptr, optr,
                                dcol, dval, ocol, oval);
    BICG_HIP(hipDeviceSynchronize());

    auto out = [&](CSR_Matrix *A, unsigned nz, unsigned cols, const unsigned *p, const unsigned *c, const double *v) {
        A->rows = rows; A->cols = cols; A->nz = nz;
        A->ptr = (unsigned *)malloc(sizeof(unsigned) * ((size_t)rows + 1));
        A->col = (unsigned *)malloc(sizeof(unsigned) * (nz ? nz : 1));
        A->val = (double *)malloc(sizeof(double) * (nz ? nz : 1));
        BICG_HIP(hipMemcpy(A->ptr, p, sizeof(unsigned) * ((size_t)rows + 1), hipMemcpyDeviceToHost));
        if (nz) {
            BICG_HIP(hipMemcpy(A->col, c, sizeof(unsigned) * nz, hipMemcpyDeviceToHost));
            BICG_HIP(hipMemcpy(A->val, v, sizeof(double) * nz, hipMemcpyDeviceToHost));
        }
    };
    out(diag, nd, rows, dptr, dcol, dval);       // cols = local rows, src/matrix.c:343-345
    out(offd, no, ncols, optr, ocol, oval);      // cols = n,          src/matrix.c:350-352
    for (void *p : {(void *)d_row, (void *)d_col, (void *)d_val, (void *)key, (void *)pos, (void *)key2, (void *)pos2, (void *)cnt_d,
                    (void *)cnt_o, (void *)dptr, (void *)optr, (void *)dcol, (void *)ocol, (void *)dval, (void *)oval, tmp})
        (void)hipFree(p);
    return 0;
}
