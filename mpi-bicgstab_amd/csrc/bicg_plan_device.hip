// bicg_plan_device.hip -- the sliced-ELL plan built ON the GPU from a device-resident CSR (bicg_create_device_csr,
// bicg_solver.cpp), and a device-side generator of the 7-point stencil of BASELINE.json configs[3] (bicg_stencil7_device).
//
// The host plan of bicg_create is one thread walking the matrix three times: ~1 s for Transport, ~5 s for a 256^3 share,
// ~40 s for the 512^3 Laplacian after 15 GB of host arrays have been generated and shipped. Here the matrix never exists
// on the host: rows are counted, scanned (rocPRIM, set-up path only) and filled in by kernels, the plan is two more
// kernels (slice lengths; the column-major padded copy) around an 8 MB prefix sum.
#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include "../../include/bicgstab_hip.h"
#include "bicg_comm.h"
#include "bicg_device.h"

namespace bicg {

namespace {

constexpr int kThreads = 256;

// slice_len[s] = longest row of slice s; *far != 0 when some |col - row| > 32767
__global__ void __launch_bounds__(kThreads) k_plan_rowstats(const uint32_t *ptr, const uint32_t *col, uint32_t rows, uint32_t *slice_len, int *far)
{
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    uint32_t len = 0;
    bool is_far = false;
    if (r < rows) {
        const uint32_t a = ptr[r], b = ptr[r + 1];
        len = b - a;
        for (uint32_t j = a; j < b; ++j) {
            const long long d = (long long)col[j] - (long long)r;
            is_far = is_far || d < -32767 || d > 32767;
        }
    }
    // the 64 rows of a slice are the 64 lanes of one wavefront
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(len, off, 64); len = o > len ? o : len; }
    if ((threadIdx.x & 63u) == 0 && r < rows) slice_len[r / kSliceRows] = len;
    if (__any(is_far) && (threadIdx.x & 63u) == 0) atomicOr(far, 1);
}

// uniform slices (SellDev::ubase): uhash[s] != 0 when all 64 rows of slice s are present, equally long, and entry k sits at
// the same distance from its row in every row; its value is a 64-bit hash of the distances (the host groups slices by hash
// and fetches one list per group). One wavefront per slice, lane = row.
__global__ void __launch_bounds__(kThreads) k_plan_uniform(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows,
                                                           unsigned long long *uhash, unsigned long long *vhash)
{
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    const bool live = r < rows;
    const uint32_t a = live ? ptr[r] : 0u, len = live ? ptr[r + 1] - a : 0u;
    const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
    bool uni = __all(live && len == len0) && len0 > 0;
    unsigned long long h = 0xcbf29ce484222325ull ^ len0;
    for (uint32_t k = 0; uni && k < len0; ++k) {
        const int d = (int)((long long)col[a + k] - (long long)r);
        const int d0 = __builtin_amdgcn_readfirstlane(d);
        uni = __all(d == d0);
        h = (h ^ (unsigned long long)(unsigned)d0) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    if ((threadIdx.x & 63u) == 0 && live) uhash[r / kSliceRows] = uni ? (h | 1ull) : 0ull;
    // ... and constant: entry k holds the same value in every row (SellDev::vbase); hash of the value bits AND the distances
    if (!vhash) return;
    bool con = uni;
    unsigned long long hv = h ^ 0x9E3779B97F4A7C15ull;
    for (uint32_t k = 0; con && k < len0; ++k) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(val[a + k]);
        const unsigned lo0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
        const unsigned long long b0 = ((unsigned long long)hi0 << 32) | lo0;
        con = __all(b == b0);
        hv = (hv ^ b0) * 0x100000001b3ull;
        hv ^= hv >> 31;
    }
    if ((threadIdx.x & 63u) == 0 && live) vhash[r / kSliceRows] = con ? (hv | 1ull) : 0ull;
}

// masked slices (SellDev::mbase): all 64 rows present, every row at most 16 entries with ascending columns, and the rows'
// (distance, value) pairs all sub-sequences of ONE ascending list of at most 16 pairs with equal values wherever a pair occurs.
// The list is built by the wavefront: the smallest distance any row still has to place becomes the next list entry, the rows
// that hold it must agree on its value and move on. Pass 1 (rmask == nullptr): mhash[s] = hash of the list, its length in the
// low 5 bits, 0 when the slice does not qualify. Pass 2: the rows' masks of the slices the host kept (mbase[s] != ~0).
__global__ void __launch_bounds__(kThreads) k_plan_masked(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows,
                                                          unsigned long long *mhash, const uint32_t *mbase, unsigned short *rmask)
{
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    const uint32_t slice = r / kSliceRows;
    const bool live = r < rows;
    if (rmask) {                                              // pass 2: only the slices that were kept
        const uint32_t first = (slice * kSliceRows < rows) ? mbase[slice] : 0xFFFFFFFFu;       // wave-uniform
        if (first == 0xFFFFFFFFu) return;
    }
    const uint32_t a = live ? ptr[r] : 0u, len = live ? ptr[r + 1] - a : 0u;
    bool ok = __all(live && len > 0 && len <= 16u);
    unsigned long long h = 0x84222325cbf29ce4ull;
    uint32_t pos = 0, mask = 0;
    int n = 0, prev = 0;
    while (ok) {
        const int cand = pos < len ? (int)((long long)col[a + pos] - (long long)r) : 0x7FFFFFFF;
        int m = cand;
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(m, off, 64); m = o < m ? o : m; }
        if (m == 0x7FFFFFFF) break;                           // every row has placed all its entries
        if (n == 16 || (n > 0 && m <= prev)) { ok = false; break; }      // list too long / a row whose columns do not ascend
        const bool has = cand == m;
        const unsigned long long vb = has ? (unsigned long long)__double_as_longlong(val[a + pos]) : 0ull;
        const unsigned long long who = __ballot(has);
        const int firstlane = __builtin_ctzll(who);
        const unsigned lo0 = (unsigned)__shfl((int)(unsigned)vb, firstlane, 64), hi0 = (unsigned)__shfl((int)(unsigned)(vb >> 32), firstlane, 64);
        const unsigned long long v0 = ((unsigned long long)hi0 << 32) | lo0;
        if (!__all(!has || vb == v0)) { ok = false; break; }  // the rows disagree on the value at this distance
        if (has) { mask |= 1u << n; ++pos; }
        h = (h ^ (unsigned long long)(unsigned)m) * 0x100000001b3ull; h ^= h >> 29;
        h = (h ^ v0) * 0x100000001b3ull; h ^= h >> 31;
        prev = m; ++n;
    }
    if (rmask) {
        if (ok && live) rmask[(size_t)(mbase[slice] & 0x03FFFFFFu) * kSliceRows + (r % kSliceRows)] = (unsigned short)mask;
        return;
    }
    if ((threadIdx.x & 63u) == 0 && live) mhash[slice] = (ok && n > 0) ? ((h << 5) | (unsigned long long)n) : 0ull;
}

// The lists the host attached to the list-driven slices are found by HASH (k_plan_uniform / k_plan_masked above; the host
// fetches one representative per hash value). A collision would hand a slice another slice's list -- a silently wrong product.
// This pass compares every row of every list-driven slice with the list it was given: distances, the values of constant /
// masked slices, the length, and for masked slices that the row's mask accounts for all its entries. bad[s] = 1 where
// anything differs; the host puts those slices back on their stored columns and values. One wavefront per slice, lane = row.
__global__ void __launch_bounds__(kThreads) k_plan_verify(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows,
                                                          const uint32_t *slice_len, const uint32_t *ubase, const uint32_t *vbase,
                                                          const uint32_t *mbase, const unsigned short *rmask, const int *uoff,
                                                          const double *uval, unsigned char *bad)
{
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    const uint32_t slice = r / kSliceRows;
    if (slice * kSliceRows >= rows) return;                                       // (wave-uniform)
    const uint32_t ub = ubase[slice];
    if (ub == 0xFFFFFFFFu) return;
    const uint32_t vb = vbase ? vbase[slice] : 0xFFFFFFFFu, mb = mbase ? mbase[slice] : 0xFFFFFFFFu;
    const bool live = r < rows;
    bool ok = live;                                                               // a list-driven slice has all its 64 rows
    if (live) {
        const uint32_t a = ptr[r], len = ptr[r + 1] - a;
        if (mb != 0xFFFFFFFFu) {
            const uint32_t ulen = mb >> 26, mask = rmask[(size_t)(mb & 0x03FFFFFFu) * kSliceRows + (r % kSliceRows)];
            uint32_t pos = 0;
            ok = vb != 0xFFFFFFFFu && ulen <= 16u && (mask >> ulen) == 0u;
            for (uint32_t k = 0; ok && k < ulen; ++k) {
                if (!((mask >> k) & 1u)) continue;
                ok = pos < len && (int)((long long)col[a + pos] - (long long)r) == uoff[ub + k] &&
                     __double_as_longlong(val[a + pos]) == __double_as_longlong(uval[vb + k]);
                ++pos;
            }
            ok = ok && pos == len;
        } else {
            ok = len == slice_len[slice];
            for (uint32_t k = 0; ok && k < len; ++k) {
                ok = (int)((long long)col[a + k] - (long long)r) == uoff[ub + k];
                if (ok && vb != 0xFFFFFFFFu) ok = __double_as_longlong(val[a + k]) == __double_as_longlong(uval[vb + k]);
            }
        }
    }
    if (!__all(ok) && (threadIdx.x & 63u) == 0) bad[slice] = 1;
}

// entry k of row r -> slice_base[r / 64] + k * 64 + r % 64 (padding stays zero); 16-bit offsets four to a word
__global__ void __launch_bounds__(kThreads) k_plan_fill(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows,
                                                        const uint32_t *slice_base, const uint32_t *slice_base16, double *sval,
                                                        uint32_t *scol, short *scol16)
{
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= rows) return;
    const uint32_t sl = r / kSliceRows, lane = r % kSliceRows;
    const size_t base = slice_base[sl];
    const size_t base16 = scol16 ? slice_base16[sl] : 0;
    const uint32_t a = ptr[r], b = ptr[r + 1];
    for (uint32_t j = a, k = 0; j < b; ++j, ++k) {
        const size_t e = base + (size_t)k * kSliceRows + lane;      // consecutive lanes write consecutive entries
        sval[e] = val[j];
        if (scol16) scol16[base16 + ((size_t)(k / 4) * kSliceRows + lane) * 4 + (k % 4)] = (short)((long long)col[j] - (long long)r);
        else scol[e] = col[j];
    }
}

// 7-point stencil on an m^3 grid, rows [lo, hi): weights = (centre, x-, x+, y-, y+, z-, z+), entries in ascending column
// order -- the matrix of mpi-bicgstab_amd/python/synth.py stencil7 (the generator bench.py and the tests use on the host)
__device__ __forceinline__ unsigned stencil_count(unsigned long long r, unsigned m)
{
    const unsigned ix = (unsigned)(r % m), iy = (unsigned)((r / m) % m), iz = (unsigned)(r / ((unsigned long long)m * m));
    return 1u + (ix > 0) + (ix < m - 1) + (iy > 0) + (iy < m - 1) + (iz > 0) + (iz < m - 1);
}
__global__ void __launch_bounds__(kThreads) k_stencil_counts(unsigned m, uint32_t rows, uint32_t *cnt)
{
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    if (r <= rows) cnt[r] = r < rows ? stencil_count(r, m) : 0u;
}
struct Weights { double w[7]; };
__global__ void __launch_bounds__(kThreads) k_stencil_fill(unsigned m, uint32_t rows, const uint32_t *ptr, Weights W, uint32_t *col, double *val)
{
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= rows) return;
    const unsigned ix = r % m, iy = (r / m) % m, iz = r / (m * m);
    uint32_t at = ptr[r];
    const uint32_t mm = m * m;
    if (iz > 0)     { col[at] = r - mm; val[at++] = W.w[5]; }
    if (iy > 0)     { col[at] = r - m;  val[at++] = W.w[3]; }
    if (ix > 0)     { col[at] = r - 1;  val[at++] = W.w[1]; }
    col[at] = r; val[at++] = W.w[0];
    if (ix < m - 1) { col[at] = r + 1;  val[at++] = W.w[2]; }
    if (iy < m - 1) { col[at] = r + m;  val[at++] = W.w[4]; }
    if (iz < m - 1) { col[at] = r + mm; val[at++] = W.w[6]; }
}

}  // namespace

void launch_plan_rowstats(const uint32_t *ptr, const uint32_t *col, uint32_t rows, uint32_t *slice_len, int *far, hipStream_t st)
{
    hipLaunchKernelGGL(k_plan_rowstats, dim3((rows + kThreads - 1) / kThreads), dim3(kThreads), 0, st, ptr, col, rows, slice_len, far);
}
void launch_plan_uniform(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows, unsigned long long *uhash,
                         unsigned long long *vhash, hipStream_t st)
{
    hipLaunchKernelGGL(k_plan_uniform, dim3((rows + kThreads - 1) / kThreads), dim3(kThreads), 0, st, ptr, col, val, rows, uhash, vhash);
}
void launch_plan_masked(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows, unsigned long long *mhash,
                        const uint32_t *mbase, unsigned short *rmask, hipStream_t st)
{
    hipLaunchKernelGGL(k_plan_masked, dim3((rows + kThreads - 1) / kThreads), dim3(kThreads), 0, st, ptr, col, val, rows, mhash, mbase, rmask);
}
void launch_plan_verify(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows, const uint32_t *slice_len,
                        const uint32_t *ubase, const uint32_t *vbase, const uint32_t *mbase, const unsigned short *rmask, const int *uoff,
                        const double *uval, unsigned char *bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_plan_verify, dim3((rows + kThreads - 1) / kThreads), dim3(kThreads), 0, st, ptr, col, val, rows, slice_len, ubase, vbase,
                       mbase, rmask, uoff, uval, bad);
}
void launch_plan_fill(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows, const uint32_t *slice_base,
                      const uint32_t *slice_base16, double *sval, uint32_t *scol, short *scol16, hipStream_t st)
{
    hipLaunchKernelGGL(k_plan_fill, dim3((rows + kThreads - 1) / kThreads), dim3(kThreads), 0, st, ptr, col, val, rows, slice_base,
                       slice_base16, sval, scol, scol16);
}

}  // namespace bicg

using namespace bicg;

// CSR of the 7-point stencil in DEVICE memory (the whole matrix: single rank). Arrays are hipMalloc'ed here; release them
// with bicg_device_free once the context has been created (bicg_create_device_csr copies what it keeps).
extern "C" int bicg_stencil7_device(unsigned int m, const double *weights, double **val_d, unsigned int **col_d, unsigned int **ptr_d,
                                    unsigned long long *nnz_out)
{
    BICG_HIP(hipSetDevice(comm_get()->device));
    const unsigned long long n64 = (unsigned long long)m * m * m, nnz64 = 7ull * n64 - 6ull * m * m;
    if (m < 2 || n64 >= 0x7FFFFFFFull || nnz64 >= 0xFFFFFF00ull) { fprintf(stderr, "ERROR: bicg_stencil7_device: grid too large for 32-bit indices\n"); return 1; }
    const uint32_t rows = (uint32_t)n64;
    uint32_t *cnt = nullptr, *ptr = nullptr, *col = nullptr;
    double *val = nullptr;
    BICG_HIP(hipMalloc((void **)&cnt, sizeof(uint32_t) * ((size_t)rows + 1)));
    BICG_HIP(hipMalloc((void **)&ptr, sizeof(uint32_t) * ((size_t)rows + 1)));
    hipLaunchKernelGGL(k_stencil_counts, dim3(rows / kThreads + 1), dim3(kThreads), 0, 0, m, rows, cnt);
    (void)hipGetLastError();
    size_t tmp_bytes = 0;
    void *tmp = nullptr;
    BICG_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt, ptr, 0u, (size_t)rows + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
    BICG_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
    BICG_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, cnt, ptr, 0u, (size_t)rows + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
    BICG_HIP(hipMalloc((void **)&col, sizeof(uint32_t) * (size_t)nnz64));
    BICG_HIP(hipMalloc((void **)&val, sizeof(double) * (size_t)nnz64));
    Weights W;
    for (int i = 0; i < 7; ++i) W.w[i] = weights[i];
    hipLaunchKernelGGL(k_stencil_fill, dim3((rows + kThreads - 1) / kThreads), dim3(kThreads), 0, 0, m, rows, ptr, W, col, val);
    BICG_HIP(hipDeviceSynchronize());
    BICG_HIP(hipFree(cnt));
    BICG_HIP(hipFree(tmp));
    *val_d = val; *col_d = col; *ptr_d = ptr;
    if (nnz_out) *nnz_out = nnz64;
    return 0;
}

extern "C" void bicg_device_free(void *p)
{
    if (p) (void)hipFree(p);
}
