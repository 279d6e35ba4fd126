// bicg_spmm.hip -- Y_j = (A + sigma_j I) X_j for up to 16 vectors with the matrix read once (the per-shift verification loop of the
// reference's shifted driver, src/test_shifted.c:129-154: BASELINE.json configs[4] "batched SpMV"), as a PIPELINE:
// k_spmm_pipe. Padded slices with 16-bit column offsets whose distances fall into clusters (struct FusedWindow: banded and
// stencil-like matrices, the Transport-shaped one among them); other layouts keep k_spmm_win / k_spmm_sell (bicg_kernels.hip).
//
// What k_spmm_win did with its 311-316 us per 16 vectors (profiles/r06/spmm_notes.txt: the kernel with parts switched off, and its
// counters): 135 us remained with no staging loads, no products and no row heads at all -- a workgroup's life was a chain of
// dependent trips (slice metadata, row heads, staging of pass 1, sigma, staging of pass 2, ...) with two workgroups per CU to
// overlap them -- and it issued 70 M vector instructions, 11.6 per entry and vector (LDS address from slot and vector, a 64-bit
// select per product for the "row has this entry" bit). Here:
//   * a step = one tile of 512 (or 256) rows x 2 vectors; two (three) workgroups per CU take turns. The x window of the NEXT step is asked for at the
//     beginning of a step (16-byte loads into registers: the clusters' runs start at even columns and even slots) and written to the OTHER of two LDS buffers at the
//     step's end, behind its products;
//   * a workgroup is persistent: an XCD owns an eighth of the groups and its workgroups take them cyclically, so that at any time
//     they work on neighbouring groups whose windows overlap in the L2 (1.28 -> 0.80 GB from the memory side per launch);
//   * the head of a group's rows (first 16 entries: all of a Transport-shaped row) stays in registers across the group's 8 steps; its
//     metadata is loaded a group ahead, by scalar loads (prefetching the heads too costs the registers of the third workgroup);
//   * a step's results are stored at the beginning of the next step; the step's shifts are requested at its beginning;
//   * an entry the row does not have is value 0.0 at the window's ZERO slot: products without predicates, one add per LDS address.
// 316 -> 175 us per 16 vectors (634 MB of matrix + X + Y: 0.45 of 8 TB/s), every column bit-identical.
// Arithmetic: per row and vector the products are added in stored order, one rounding per product and per sum, y = 0 + that sum,
// then the offd part, then sigma_j x_j -- bit for bit k_spmm_win's, i.e. bicg_spmv's column by column (tests/test_full_size.py,
// tests/test_shifted.py, across ranks tests/test_multirank.py).
#include "bicg_device.h"
#include <hip/hip_ext.h>
#include "bicg_devfn.h"
#include "bicg_reduce.h"
#include "bicg_knobs.h"

#include <cstdio>
#include <cstdlib>

namespace bicg {

extern __shared__ double spmm_lds[];
typedef short spmm_i16x4 __attribute__((ext_vector_type(4)));

// Shape of a step (compile-time; measured with 16 vectors on the Transport-shaped matrix, kernel time by rocprofv3,
// profiles/r06/spmm_notes.txt): vectors per step x resident workgroups per CU x next group's row heads prefetched or not
//   4 x 2 x yes  203-206 us (225 registers)      4 x 2 x no  208 us
//   2 x 3 x no   184-186 us (145 registers, 40 KB of LDS per workgroup)   <- 256-row tiles
//   2 x 2 workgroups of 512 rows: 175 us (128 registers + 20 bytes of scratch, 64 KB of LDS)
//   ... and the next group's metadata asked for after the last step instead of in front of the first (SPMM_PREFETCH_META 0: 127
//   registers, no scratch): 155-156 us   <- the default where the window fits
//   2 x 3 x yes  spills (168 registers + 60 bytes)      2 x 4 x no  218 us (128 registers + 60 bytes of scratch)      1 x 4 x no  219 us
// Smaller steps let a third workgroup per CU take turns with the other two; without the prefetch the heads of a group's rows are
// waited for once per 8 steps.
#ifndef SPMM_NV
#define SPMM_NV 2
#endif
#ifndef SPMM_RESIDENT
#define SPMM_RESIDENT 768
#endif
#ifndef SPMM_WAVES
#define SPMM_WAVES 3
#endif
#ifndef SPMM_PREFETCH_HEAD
#define SPMM_PREFETCH_HEAD 0
#endif
#ifndef SPMM_PREFETCH_META
#define SPMM_PREFETCH_META 0
#endif
#ifndef SPMM_RESIDENT512
#define SPMM_RESIDENT512 512
#endif
constexpr int kDmaNV = SPMM_NV;    // vectors per step: two buffers of 2 x 1 248 slots = 40 KB, three workgroups per CU
constexpr int kDmaHead = 16;       // entries of a row kept in registers across the passes of its group

// what a group's rows need before their entries can be asked for (loaded a whole group ahead) ...
#define BICG_KCONST __attribute__((address_space(4)))
struct SpmmMeta { uint32_t p0, p1, len, base, base16, oa, ob; double bi; };      // (p1 - p0: the row's length -- subtracted where it is used)
// ... and the head of the rows: the first 16 entries (value, LDS slot, "counts" bit)
struct SpmmHead { double v[kDmaHead]; unsigned s8[kDmaHead]; };      // s8: BYTE offset of the entry's slot in a vector's window

// LDS slot of the entry at distance d from the row of thread tid (padding: distance 0, the row's own column)
__device__ __forceinline__ unsigned dma_slot(const FusedWindow &cl, unsigned tid, int d)
{
    int bias = cl.bias[0];
    if (cl.ncl > 1 && d >= cl.lo[1]) bias = cl.bias[1];
    if (cl.ncl > 2 && d >= cl.lo[2]) bias = cl.bias[2];
    if (cl.ncl > 3 && d >= cl.lo[3]) bias = cl.bias[3];
    return (unsigned)((int)tid + d + bias);
}

template <bool OFFD, int TR>
__device__ __forceinline__ void dma_meta(const SpmmArgs &a, unsigned g, unsigned tid, unsigned wave, SpmmMeta &M)
{
    const uint32_t row = g * (uint32_t)TR + tid;
    const uint32_t slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)(g * (uint32_t)(TR / kSliceRows) + wave));     // scalar loads below
    const bool live = row < a.nrows;
    M.base = 0u; M.len = 0u; M.base16 = 0u;
    if (slice * kSliceRows < a.nrows) {       // (scalar loads: they do not take part in the vector memory counter)
        M.base = *((const BICG_KCONST uint32_t *)a.sell.slice_base + slice);
        M.len = *((const BICG_KCONST uint32_t *)a.sell.slice_len + slice);
        M.base16 = *((const BICG_KCONST uint32_t *)a.sell.slice_base16 + slice);
    }
    M.p0 = 0u; M.p1 = 0u;
    if (live) { M.p0 = a.dptr[row]; M.p1 = a.dptr[row + 1]; }
    M.oa = 0u; M.ob = 0u;
    if (OFFD && live) { M.oa = a.offd.ptr[row]; M.ob = a.offd.ptr[row + 1]; }
    M.bi = (a.b && live) ? a.b[row] : 0.0;
}

// the raw head of a group's rows as it comes from memory: NOTHING is computed from it where it is requested (an instruction that
// uses a loaded word makes the wavefront wait for it -- and, the counter being in order, for everything asked for before it)
struct SpmmRaw { double v[kDmaHead]; spmm_i16x4 q[kDmaHead / 4]; };
__device__ __forceinline__ void dma_vals(const SpmmArgs &a, const SpmmMeta &M, unsigned lane, SpmmRaw &R)
{
    const spmm_i16x4 *const q16 = reinterpret_cast<const spmm_i16x4 *>(a.sell.col16) + ((size_t)M.base16 / 4 + lane);
#pragma unroll
    for (int q = 0; q < kDmaHead / 4; ++q) {
        R.q[q] = (spmm_i16x4)(0);
        if ((uint32_t)(4 * q) < M.len) R.q[q] = q16[(size_t)q * kSliceRows];       // wave-uniform test; the quad is padded
    }
#pragma unroll
    for (int e = 0; e < kDmaHead; ++e) R.v[e] = (uint32_t)e < M.len ? a.sell.val[M.base + (uint32_t)e * kSliceRows + lane] : 0.0;
}
// An entry the row does not have (padding of the slice, or past the head of a shorter row) becomes value 0.0 at the window's ZERO
// slot (the last slot of every vector's window holds 0.0): the products need no per-entry predicate then -- acc + 0.0 * 0.0 is
// acc (a sum of -0.0 would turn +0.0: y = 0.0 + sum gives +0.0 either way, reference src/matrix.c:434-437).
__device__ __forceinline__ void dma_finish(const SpmmArgs &a, const SpmmMeta &M, const SpmmRaw &R, unsigned tid, SpmmHead &H)
{
    const unsigned zero8 = (a.wslots - 1u) * 8u, mylen = M.p1 - M.p0;
#pragma unroll
    for (int q = 0; q < kDmaHead / 4; ++q) {
        H.s8[4 * q + 0] = dma_slot(a.cl, tid, R.q[q].x) * 8u; H.s8[4 * q + 1] = dma_slot(a.cl, tid, R.q[q].y) * 8u;
        H.s8[4 * q + 2] = dma_slot(a.cl, tid, R.q[q].z) * 8u; H.s8[4 * q + 3] = dma_slot(a.cl, tid, R.q[q].w) * 8u;
    }
#pragma unroll
    for (int e = 0; e < kDmaHead; ++e) {
        const bool on = (uint32_t)e < mylen;
        H.v[e] = on ? R.v[e] : 0.0;
        H.s8[e] = on ? H.s8[e] : zero8;
    }
}

// The window of group g for vectors v0 .. v0 + 3, through registers: thread t owns the 16-byte pairs t, t + 256, t + 512 of the window
// (the clusters' runs lie back to back in LDS, start at even columns and hold even numbers of columns: pair f is slots 2 f, 2 f + 1
// whatever its cluster), asks for them at the beginning of a step and writes them to the other buffer at its end -- the loads
// have the step's products to land behind. (The same copies through the DMA path, global_load_lds 16 bytes per lane, no staging
// registers: 243 us per launch against 203 -- that path delivered 16 GB/s per CU here, 4.1 TB/s chip-wide for the gigabyte of
// windows a launch stages. profiles/r06/spmm_notes.txt)
template <int TR> struct StageShape { static constexpr int J = TR == 256 ? 3 : 2; };      // pairs per thread and vector: windows of up to 1 536 (2 048) slots
typedef double spmm_f64x2 __attribute__((ext_vector_type(2)));
template <int TR> struct SpmmStage { spmm_f64x2 t[StageShape<TR>::J][kDmaNV]; };
template <int TR>
__device__ __forceinline__ void stage_load(const SpmmArgs &a, unsigned g, int v0, unsigned tid, SpmmStage<TR> &T)
{
    const int g0 = (int)(g * (unsigned)TR), last = (int)a.nrows;
    const int n0 = (TR + a.cl.hi[0] - a.cl.lo[0]) / 2;
    const int n1 = a.cl.ncl > 1 ? n0 + (TR + a.cl.hi[1] - a.cl.lo[1]) / 2 : n0;
    const int n2 = a.cl.ncl > 2 ? n1 + (TR + a.cl.hi[2] - a.cl.lo[2]) / 2 : n1;
    const int n3 = a.cl.ncl > 3 ? n2 + (TR + a.cl.hi[3] - a.cl.lo[3]) / 2 : n2;
#pragma unroll
    for (int j = 0; j < StageShape<TR>::J; ++j) {
        const int f = (int)tid + j * TR;
        // the pair's first column: cluster by position, distance = lo_k + 2 (f - pairs before the cluster)
        int d = a.cl.lo[0] + 2 * f;
        if (f >= n0) d = a.cl.lo[1] + 2 * (f - n0);
        if (f >= n1) d = a.cl.lo[2] + 2 * (f - n1);
        if (f >= n2) d = a.cl.lo[3] + 2 * (f - n2);
        const int col = g0 + d;
        const bool ok = f < n3 && col >= 0 && col < last;
#pragma unroll
        for (int v = 0; v < kDmaNV; ++v) {
            T.t[j][v] = (spmm_f64x2)(0.0);
            if (ok && v0 + v < a.nvec) T.t[j][v] = *reinterpret_cast<const spmm_f64x2 *>(a.xs + (size_t)(v0 + v) * a.vstride + col);
        }
    }
}
template <int TR>
__device__ __forceinline__ void stage_store(const SpmmArgs &a, double *dst, unsigned tid, const SpmmStage<TR> &T)
{
    const unsigned W = a.wslots, pairs = (W - 2u) / 2u;
#pragma unroll
    for (int j = 0; j < StageShape<TR>::J; ++j) {
        const unsigned f = tid + (unsigned)j * (unsigned)TR;
        if (f < pairs) {
#pragma unroll
            for (int v = 0; v < kDmaNV; ++v) *reinterpret_cast<spmm_f64x2 *>(dst + (size_t)v * W + 2u * f) = T.t[j][v];
        }
    }
}

template <bool OFFD, int TR>
__global__ void __launch_bounds__(TR) __attribute__((amdgpu_waves_per_eu(TR == 512 ? 4 : SPMM_WAVES, 4))) k_spmm_pipe(SpmmArgs a)
{
    constexpr int NV = kDmaNV, K = kDmaHead, U = 8;
    __shared__ double sm[(TR / 64) * NV];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned W = a.wslots;
    double *const buf0 = spmm_lds, *const buf1 = spmm_lds + (size_t)NV * W;
    // Which groups: workgroup b runs on XCD b % 8. gstep > 0: consecutive groups floor(vb gstep) ..., XCD-contiguous ranges.
    // gstep == 0 (default): every XCD owns an eighth of the groups and its workgroups take them CYCLICALLY -- at any time the
    // workgroups of an XCD work on neighbouring groups, whose windows overlap: what one of them fetched the others find in the L2.
    const unsigned nwg = gridDim.x;
    unsigned gfirst, gend, gstride;
    if (a.gstep > 0.0) {
        unsigned vb = blockIdx.x;
        if (a.xcd_map && nwg % 8u == 0u) vb = (blockIdx.x % 8u) * (nwg / 8u) + blockIdx.x / 8u;
        gfirst = (unsigned)((double)vb * a.gstep);
        const double upto = (double)(vb + 1u) * a.gstep;
        gend = upto >= (double)a.ngroups ? a.ngroups : (unsigned)upto;      // (the grid is padded to a multiple of 8: the last few own nothing)
        gstride = 1u;
    } else {
        const unsigned per = (a.ngroups + 7u) / 8u, xcd = blockIdx.x % 8u, j = blockIdx.x / 8u;
        gstride = nwg / 8u;
        gfirst = xcd * per + j;
        gend = (xcd + 1u) * per < a.ngroups ? (xcd + 1u) * per : a.ngroups;
    }
    const int npass = (a.nvec + NV - 1) / NV;
    if (a.b) {      // columns past the last vector (rows past the last group: launch_spmm_pipe)
        for (unsigned g = gfirst; g < gend; g += gstride)
            if (tid < (unsigned)kSpmmCols && (int)tid >= a.nvec) a.partial[(size_t)g * kSpmmCols + tid] = 0.0;
    }
    if (gfirst >= gend) return;

    // In-order memory counter: whatever a step WAITS for is asked for before the step's window loads, or the wait covers those
    // too. A group's metadata is loaded a group ahead (pass 0, scalar loads and one row-pointer pair), its row heads in the last
    // step of the group before (in front of the window loads; converted behind that step's hand-over barrier).
    // (Taking the groups in pairs one cluster distance apart, passes outermost inside a pair -- the far window of one group is the
    // near window of the other one step later -- lowered the L2 misses from 0.80 to 0.68 GB and left the time where it was, at the
    // price of a second set of row heads in registers: not kept. profiles/r06/spmm_notes.txt)
    if (tid < 2u * NV) spmm_lds[(size_t)tid * W + (W - 1u)] = 0.0;      // the ZERO slot of every vector's window, both buffers (dma_finish)
    SpmmMeta M0, MN0;
    SpmmHead H0;
    SpmmRaw N;
    dma_meta<OFFD, TR>(a, gfirst, tid, wave, M0);
    dma_vals(a, M0, lane, N);
    SpmmStage<TR> T;
    stage_load<TR>(a, gfirst, 0, tid, T);
    stage_store<TR>(a, buf0, tid, T);
    dma_finish(a, M0, N, tid, H0);
    MN0 = M0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    double yprev[NV];
    uint32_t prev_row = 0xFFFFFFFFu;
    int prev_v0 = 0;
    unsigned step = 0;
    // one step: group g, pass p out of the current buffer; (ng, np): the step after it (ng = ~0u: none); rawm: the metadata of the
    // group whose row heads are to be requested in this step (want_raw)
    auto run = [&](SpmmHead &H, const SpmmMeta M, unsigned g, int p, unsigned ng, int np, bool want_raw, const SpmmMeta rawm) {
        const uint32_t row = g * (uint32_t)TR + tid;
        const bool live = row < a.nrows;
        const int v0 = p * NV, nv = a.nvec - v0 < NV ? a.nvec - v0 : NV;
        double *const cur = (step & 1u) ? buf1 : buf0, *const nxt = (step & 1u) ? buf0 : buf1;
        ++step;
        double sg[NV];                                                // the step's shifts: requested now, used behind the products
#pragma unroll
        for (int v = 0; v < NV; ++v) sg[v] = (a.sigma && v < nv) ? a.sigma[v0 + v] : 0.0;
        // ---- what later steps need: row heads (raw), then the next step's window
        if (want_raw) dma_vals(a, rawm, lane, N);
        if (ng != 0xFFFFFFFFu) stage_load<TR>(a, ng, np * NV, tid, T);
        // ---- the previous step's results
        if (prev_row != 0xFFFFFFFFu && a.ys) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (prev_v0 + v < a.nvec) a.ys[(size_t)(prev_v0 + v) * a.vstride + prev_row] = yprev[v];
        }
        // ---- this step: NV sums per lane out of the current buffer, the head from registers, whatever follows streamed
        double acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.0;
#pragma unroll
        for (int e = 0; e < K; ++e) asm volatile("" : "+v"(H.s8[e]));     // (keeps the 16 x NV LDS addresses out of loop-invariant registers)
        const char *const cb = reinterpret_cast<const char *>(cur);
        auto half = [&](int e0) {       // four entries' reads (16) in flight, then their products in stored order
#pragma unroll
            for (int e4 = e0; e4 < e0 + K / 2; e4 += 4) {
                double xr[4][NV];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int v = 0; v < NV; ++v) xr[i][v] = *reinterpret_cast<const double *>(cb + (size_t)v * W * 8u + H.s8[e4 + i]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v] = acc[v] + H.v[e4 + i] * xr[i][v];       // an absent entry adds 0.0 * 0.0
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        half(0);
        if (M.len > (uint32_t)(K / 2)) half(K / 2);
        for (uint32_t k0 = K; k0 < M.len; k0 += U) {                  // rows longer than the head: streamed per pass
            const spmm_i16x4 *const q16 = reinterpret_cast<const spmm_i16x4 *>(a.sell.col16) + ((size_t)M.base16 / 4 + lane);
            double val[U];
            unsigned sl[U];
#pragma unroll
            for (int q = 0; q < U / 4; ++q) {
                spmm_i16x4 dq = (spmm_i16x4)(0);
                if (k0 + 4 * q < M.len) dq = q16[(size_t)(k0 / 4 + q) * kSliceRows];
                sl[4 * q + 0] = dma_slot(a.cl, tid, dq.x); sl[4 * q + 1] = dma_slot(a.cl, tid, dq.y);
                sl[4 * q + 2] = dma_slot(a.cl, tid, dq.z); sl[4 * q + 3] = dma_slot(a.cl, tid, dq.w);
            }
#pragma unroll
            for (int e = 0; e < U; ++e) val[e] = k0 + e < M.len ? a.sell.val[M.base + (k0 + e) * kSliceRows + lane] : 0.0;
#pragma unroll
            for (int e = 0; e < U; ++e) {
                const bool on = k0 + e < M.p1 - M.p0;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double t = acc[v] + val[e] * cur[(unsigned)v * W + sl[e]];
                    acc[v] = on ? t : acc[v];
                }
            }
        }
        double r2[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            r2[v] = 0.0;
            yprev[v] = 0.0;
            if (v < nv && live) {
                double y = 0.0 + acc[v];                              // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
                if (OFFD) {
                    const double *xv = a.xs + (size_t)(v0 + v) * a.vstride;
                    double so = 0.0;
                    for (uint32_t k = M.oa; k < M.ob; ++k) so += a.offd.val[k] * xv[a.offd.col[k]];
                    y += so;                                          // second mult() call, src/matrix.c:440
                }
                // += sigma_j x_j (src/test_shifted.c:133): the row's own column is in the window (distance 0 belongs to a cluster)
                if (a.sigma) y += sg[v] * cur[(unsigned)v * W + dma_slot(a.cl, tid, 0)];
                yprev[v] = y;
                if (a.b) { const double dd = (M.bi + (-1.0) * y) - 0.0; r2[v] = dd * dd; }
            }
        }
        prev_row = live ? row : 0xFFFFFFFFu; prev_v0 = v0;
        if (a.b) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double t = wave_sum(r2[v]);
                if (lane == 0) sm[wave * NV + v] = t;
            }
            __syncthreads();
            if ((int)tid < nv) {
                double t = sm[tid];
                for (int w = 1; w < TR / 64; ++w) t += sm[w * NV + tid];
                a.partial[(size_t)g * kSpmmCols + v0 + tid] = t;
            }
        }
        // ---- hand-over: the next step's window has landed, nobody reads this step's buffer any more
        if (ng != 0xFFFFFFFFu) stage_store<TR>(a, nxt, tid, T);
        __syncthreads();
    };

    const unsigned none = 0xFFFFFFFFu;
    for (unsigned g = gfirst; g < gend; g += gstride) {
        const bool more = g + gstride < gend;
        for (int p = 0; p < npass; ++p) {
            const bool last = p + 1 == npass;
            if (SPMM_PREFETCH_META && p == 0 && more) dma_meta<OFFD, TR>(a, g + gstride, tid, wave, MN0);       // the next group's metadata, in front of everything this step asks for
            run(H0, M0, g, p, !last ? g : (more ? g + gstride : none), !last ? p + 1 : 0, SPMM_PREFETCH_HEAD && last && more, MN0);
        }
        if (more) {
            if (!SPMM_PREFETCH_META) dma_meta<OFFD, TR>(a, g + gstride, tid, wave, MN0);
            M0 = MN0;
            if (!SPMM_PREFETCH_HEAD) dma_vals(a, M0, lane, N);       // (waited for right here: once per group)
            dma_finish(a, M0, N, tid, H0);
        }
    }
    if (prev_row != 0xFFFFFFFFu && a.ys) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (prev_v0 + v < a.nvec) a.ys[(size_t)(prev_v0 + v) * a.vstride + prev_row] = yprev[v];
    }
}

// Padded slices with 16-bit offsets in clusters (col16 is complete: uniform / constant lists are not needed), and a window that
// fits two buffers of 4 vectors and the pairs a thread stages. The clusters are re-laid for 16-byte copies: every run starts at an
// even distance and an even slot and holds an even number of columns. `tile` rows per workgroup (= its threads): 256 or 512.
bool spmm_pipe_plan(const SpmmArgs &a, int tile, FusedWindow &out, unsigned &wslots)
{
    if (a.cl.ncl <= 0 || a.sell.jag || a.sell.win_slots || !a.sell.col16 || !a.sell.slice_base16) return false;
    FusedWindow f = a.cl;
    int slots = 0;
    for (int k = 0; k < f.ncl; ++k) {
        // the run of cluster k covers the distances lo .. hi for rows 0 .. tile - 1 of the workgroup: columns g0 + lo .. g0 + tile - 1
        // + hi, i.e. tile + hi - lo of them -- even when lo and hi are
        f.lo[k] = a.cl.lo[k] & ~1;                       // rounds towards minus infinity (two's complement)
        f.hi[k] = (a.cl.hi[k] + 1) & ~1;
        if (k > 0 && f.lo[k] <= f.hi[k - 1] + tile) return false;
        f.bias[k] = slots - f.lo[k];
        slots += tile + f.hi[k] - f.lo[k];
    }
    if (slots > 2 * (tile == 256 ? 3 : 2) * tile) return false;      // the 16-byte pairs a thread stages per vector (StageShape)
    slots += 2;                                          // the last slot of a window holds 0.0 (dma_finish); even count
    if ((size_t)2 * kDmaNV * (size_t)slots * 8u > 158u * 1024u) return false;
    f.slots = (unsigned)slots;
    out = f; wslots = (unsigned)slots;
    return true;
}

// Resident workgroups: three per CU with 256-row tiles (40 KB of LDS, 145 registers each), one with 512-row tiles: all of them from the
// start, an XCD's workgroups taking its tiles cyclically. (Marching through consecutive groups in step with a workgroup one cluster
// distance ahead was tried and measured no gain: profiles/r06/spmm_notes.txt.)
static void spmm_pipe_shape(const SpmmArgs &a, unsigned ntiles, unsigned resident, unsigned &grid, double &gstep)
{
    grid = ntiles <= resident ? ((ntiles + 7u) & ~7u) : resident;
    gstep = 0.0;                                                               // cyclic within the XCD's eighth
    if (const char *v = test_tok("spmm-gstep")) {                              // (measurement: consecutive tiles per workgroup, in thousandths)
        const double s = 1e-3 * atof(v), ng = (double)ntiles;
        if (s >= 1.0) {
            unsigned g = (unsigned)(ng / s) + 1u;
            while ((double)(g - 1u) * s >= ng) --g;
            grid = (g + 7u) & ~7u; gstep = s;
        }
    }
}

hipError_t launch_spmm_pipe(const SpmmArgs &a0, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    if (a0.ngroups == 0) return hipSuccess;
    SpmmArgs a = a0;
    FusedWindow f;
    unsigned W = 0;
    // 512-row tiles stage 3.9 x the vectors instead of 4.9 x (the clusters' spans are shared by twice the rows); with two vectors per
    // step two such workgroups (64 KB of LDS, 128 registers, 8 wavefronts each) are resident per CU: 175 us against 185 for three
    // 256-row workgroups. BICG_TEST="spmm-tile=256" selects the latter; a window too long for 512-row tiles falls back to it.
    int tile = 512;
    if (const char *v = test_tok("spmm-tile")) tile = atoi(v) == 256 ? 256 : 512;
    if (tile == 512 && !spmm_pipe_plan(a0, 512, f, W)) tile = 256;
    if (tile == 256 && !spmm_pipe_plan(a0, 256, f, W)) return hipErrorInvalidValue;
    a.cl = f; a.wslots = W;
    const unsigned ntiles = (a0.nrows + (unsigned)tile - 1u) / (unsigned)tile, ngroups_all = a0.ngroups;
    a.ngroups = ntiles;                                                         // the kernel counts tiles
    unsigned grid = 0;
    spmm_pipe_shape(a, ntiles, tile == 256 ? (unsigned)SPMM_RESIDENT : (unsigned)SPMM_RESIDENT512, grid, a.gstep);
    const unsigned lds = 2u * (unsigned)kDmaNV * W * 8u;
    // one row of partial sums per TILE (not per workgroup); the column sums run over spmm_grid(groups) rows: those beyond the last tile are zero
    if (a.b) (void)hipMemsetAsync(a.partial + (size_t)ntiles * kSpmmCols, 0, sizeof(double) * ((size_t)ngroups_all + 8 - ntiles) * kSpmmCols, st);
    auto go = [&](auto kernel) {
        static bool raised = false;       // (one flag per instantiation of this lambda's call operator: the attribute call is a host round trip)
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            (void)hipGetLastError();
            raised = true;
        }
        if (e0 && e1) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3((unsigned)tile), lds, st, e0, e1, 0, a);
        else hipLaunchKernelGGL(kernel, dim3(grid), dim3((unsigned)tile), lds, st, a);
        return hipGetLastError();
    };
    if (tile == 512) return with_offd ? go(k_spmm_pipe<true, 512>) : go(k_spmm_pipe<false, 512>);
    return with_offd ? go(k_spmm_pipe<true, 256>) : go(k_spmm_pipe<false, 256>);
}

void preload_spmm_kernels()
{
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k_spmm_pipe<false, 512>));
    (void)hipGetLastError();
}

}  // namespace bicg
