// bicg_spmm_jag.hip -- Y_j = (A + sigma_j I) X_j for up to 16 vectors with the matrix read once, for RAGGED rows: k_spmm_jpipe,
// the pipeline of k_spmm_pipe (bicg_spmm.hip) on jagged slices whose 16-bit words are LDS slots of the group's x window
// (struct SellDev: jag, win_runs / win_list, perm, lane_info) -- the layouts of the unstructured mesh matrix in its RCM and
// generator numberings and of the FEM-like matrix. The reference multiplies A with many vectors in the verification loop of its
// shifted driver, one product per shift (src/test_shifted.c:129-154: BASELINE.json configs[4] "batched SpMV").
//
// k_spmm_win (bicg_kernels.hip) walked these layouts as one workgroup per 256-row group, eight vectors per pass, every pass a chain
// of dependent trips (columns of the window -> 8 x values per slot -> LDS -> products, and one more trip per eight entries behind
// a row's 16th) with two workgroups per CU to overlap them: 380 / 426 / 498 us for 16 vectors on the FEM-like matrix and the mesh in
// generator / RCM order. Here (what each part cost before it was changed: profiles/r06/spmm_jag_notes.txt):
//   * a step = one 256-row group x 2 vectors; workgroups are persistent, an XCD's workgroups take its groups cyclically;
//   * the columns of the slots a thread stages (slot t + 256 j) come ONCE per group -- the words of the group's column list, or
//     16-bit distances computed from its run descriptors -- a group ahead, and stay in registers for the group's eight steps;
//   * the x values of the NEXT step are asked for at the beginning of a step and written to LDS at its end, behind the products and
//     a barrier (one window buffer: three workgroups per CU);
//   * the first 16 entries of a row stay in registers across the group's steps (requested raw: nothing is computed from a loaded
//     word where it is asked for), the entries BEHIND them are copied to LDS once per group -- a jagged slice stores them
//     back to back -- and read from there in every step (from memory: one dependent trip per four entries and step, 110 us per
//     launch); an entry a row does not have is value 0.0 at the window's ZERO slot (no predicates);
//   * a step's results go through LDS (the lanes hold the group's rows by decreasing length) and are stored row by row at the
//     beginning of the next step.
// Arithmetic: per row and vector the products are added in stored order, one rounding per product and per sum, y = 0 + that sum,
// then the offd part, then sigma_j x_j -- bit for bit k_spmm_win's, i.e. bicg_spmv's column by column (tests/test_mesh_gpu.py,
// tests/test_shifted.py, tests/test_full_size.py).
#include "bicg_device.h"
#include <hip/hip_ext.h>
#include "bicg_devfn.h"
#include "bicg_reduce.h"
#include "bicg_knobs.h"

#include <cstdio>
#include <cstdlib>

namespace bicg {

extern __shared__ double spmm_lds[];

#ifndef JPIPE_WAVES
#define JPIPE_WAVES 3
#endif
// (measured, tools/jpipe_variants.sh: eight entries per trip behind the heads 342 against 331 us, eight LDS reads in flight 334, two
// wavefronts per SIMD without spills 401; groups drawn from a counter per XCD instead of dealt 340 against 333)
#ifndef JPIPE_U
#define JPIPE_U 4
#endif
#ifndef JPIPE_XR
#define JPIPE_XR 4
#endif
constexpr int kJpNV = 2;          // vectors per step
constexpr int kJpHead = 16;       // entries of a row kept in registers across the steps of its group
#define BICG_KCONST __attribute__((address_space(4)))

struct JpMeta { uint32_t base, len, row, mylen, oa, ob; double bi; bool live; };
struct JpHead { double v[kJpHead]; unsigned s8[kJpHead]; };      // s8: BYTE offset of the entry's slot in a vector's window

// What a group switch needs FIRST -- the slices' base and length, the lanes' rows and lengths -- is requested a whole group ahead
// (three dependent trips per switch otherwise: these words, then the heads they locate; 11 us per group and workgroup)
struct JpAhead { uint32_t base, len; unsigned li; };
__device__ __forceinline__ void jp_ahead(const SpmmArgs &a, unsigned g, unsigned tid, unsigned wave, JpAhead &A)
{
    const uint32_t g0 = g * (uint32_t)kGroupRows;
    const uint32_t slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)(g * (uint32_t)(kGroupRows / kSliceRows) + wave));
    A.base = 0u; A.len = 0u;
    if (slice * kSliceRows < a.nrows) {       // scalar loads
        A.base = *((const BICG_KCONST uint32_t *)a.sell.slice_base + slice);
        A.len = *((const BICG_KCONST uint32_t *)a.sell.slice_len + slice);
    }
    A.li = a.sell.lane_info[(size_t)g0 + tid];      // row within the group (low byte), its length (high byte)
}
template <bool OFFD>
__device__ __forceinline__ void jp_meta(const SpmmArgs &a, unsigned g, const JpAhead &A, JpMeta &M)
{
    const uint32_t g0 = g * (uint32_t)kGroupRows;
    M.base = A.base; M.len = A.len;
    M.row = g0 + (A.li & 0xFFu);
    M.live = M.row < a.nrows;
    M.mylen = M.live ? A.li >> 8 : 0u;
    M.oa = 0u; M.ob = 0u;
    if (OFFD && M.live) { M.oa = a.offd.ptr[M.row]; M.ob = a.offd.ptr[M.row + 1]; }
    M.bi = (a.b && M.live) ? a.b[M.row] : 0.0;
}

// The head of the group's rows as it comes from memory (step e of a jagged slice stores the entries of the rows longer than e, in
// lane order). NOTHING is computed from a loaded word here: an instruction that uses one makes the wavefront wait for it -- 16
// dependent trips per group when the slot was scaled where it was loaded (80 us per launch).
__device__ __forceinline__ uint32_t jp_head_request(const SpmmArgs &a, const JpMeta &M, JpHead &H)
{
    const unsigned short *const slots16 = reinterpret_cast<const unsigned short *>(a.sell.col16);
    uint32_t pos = M.base;
#pragma unroll
    for (int e = 0; e < kJpHead; ++e) {
        H.s8[e] = 0u; H.v[e] = 0.0;
        if ((uint32_t)e < M.len) {                            // wave-uniform
            const bool mine = (uint32_t)e < M.mylen;
            const unsigned long long m = __ballot(mine);
            const uint32_t j = pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            pos += (uint32_t)__builtin_popcountll(m);
            if (mine) { H.s8[e] = slots16[j]; H.v[e] = a.sell.val[j]; }
        }
    }
    return pos;       // first entry of step kJpHead
}
__device__ __forceinline__ void jp_head_finish(const JpMeta &M, unsigned zero8, JpHead &H)
{
#pragma unroll
    for (int e = 0; e < kJpHead; ++e) H.s8[e] = (uint32_t)e < M.mylen ? H.s8[e] * 8u : zero8;
}

// The columns of the slots thread tid stages for group g (slot tid + 256 j), as 16-bit distances from the group's first row, two per
// word: word h holds slots tid + 512 h (low half) and tid + 512 h + 256. A slot nobody reads (past the group's last) has distance 0:
// its loads need no predicate, they fetch x of the group's first row.
template <int JJ>
__device__ __forceinline__ void jp_cols(const SpmmArgs &a, unsigned g, unsigned tid, uint2 *wruns, uint32_t (&cw)[(JJ + 1) / 2])
{
    if (a.sell.win_list) {      // the list IS these words (bicg_create.cpp, SellDev::win_list)
        const uint32_t lbase = a.sell.win_lptr[g], ltotal = a.sell.win_ltotal[g];
#pragma unroll
        for (int h = 0; h < (JJ + 1) / 2; ++h) {
            cw[h] = 0u;
            if (512u * (unsigned)h < ltotal) cw[h] = a.sell.win_list[lbase + (uint32_t)h * (uint32_t)kGroupRows + tid];
        }
    } else {
        const int g0 = (int)(g * (unsigned)kGroupRows);
        const uint32_t r0 = a.sell.win_ptr[g];
        const unsigned nwr = a.sell.win_ptr[g + 1] - r0;
        __syncthreads();                                      // nobody searches the previous group's runs any more
        if (tid < nwr && tid < 64u) wruns[tid] = a.sell.win_runs[r0 + tid];
        __syncthreads();
        unsigned r = 0;
#pragma unroll
        for (int h = 0; h < (JJ + 1) / 2; ++h) cw[h] = 0u;
#pragma unroll
        for (int j = 0; j < JJ; ++j) {
            const unsigned sl = tid + 256u * (unsigned)j;
            while (r + 1 < nwr && (wruns[r + 1].y >> 16) <= sl) ++r;      // runs are ordered by slot
            const unsigned off = sl - (wruns[r].y >> 16);
            const uint32_t d = (nwr && off < (wruns[r].y & 0xFFFFu)) ? (uint32_t)((int)(wruns[r].x + off) - g0) & 0xFFFFu : 0u;
            cw[j / 2] |= (j & 1) ? d << 16 : d;
        }
    }
}

template <int JJ> struct JpStage { double t[JJ][kJpNV]; };
template <int JJ>
__device__ __forceinline__ void jp_stage_load(const SpmmArgs &a, unsigned g, const uint32_t (&cw)[(JJ + 1) / 2], int v0, JpStage<JJ> &T)
{
    const int g0 = (int)(g * (unsigned)kGroupRows);
#pragma unroll
    for (int v = 0; v < kJpNV; ++v) {
        const int vv = v0 + v < a.nvec ? v0 + v : a.nvec - 1;       // (a vector past the last: read again, never used)
        const double *const xv = a.xs + (size_t)vv * a.vstride;
#pragma unroll
        for (int j = 0; j < JJ; ++j) {
            uint32_t w = cw[j / 2];
            asm volatile("" : "+v"(w));      // (decoded per step: the byte offsets of a group's slots are not worth seven registers)
            const int d = (int)(short)((j & 1) ? w >> 16 : w & 0xFFFFu);
            // (32-bit BYTE offset from a uniform base: one address register per load instead of two; spmm_possible bounds the stride)
            T.t[j][v] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(xv) + (unsigned)(g0 + d) * 8u);
        }
    }
}
template <int JJ>
__device__ __forceinline__ void jp_stage_store(const SpmmArgs &a, double *dst, unsigned tid, const JpStage<JJ> &T)
{
    const unsigned W = a.wslots;
#pragma unroll
    for (int j = 0; j < JJ; ++j) {
        const unsigned sl = tid + 256u * (unsigned)j;
        if (sl + 1u < W) {                                    // (the last slot is the ZERO slot)
#pragma unroll
            for (int v = 0; v < kJpNV; ++v) dst[(size_t)v * W + sl] = T.t[j][v];
        }
    }
}

template <bool OFFD, int JJ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(JPIPE_WAVES, 4))) k_spmm_jpipe(SpmmArgs a, unsigned tq)
{
    constexpr int NV = kJpNV, K = kJpHead, U = JPIPE_U, XR = JPIPE_XR, NW = kGroupRows / 64;
    __shared__ double sm[NW * NV];
    __shared__ uint2 wruns[64];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned W = a.wslots, zero8 = (W - 1u) * 8u;
    // LDS: the window of NV vectors | the slices' entries behind their rows' heads (values) | two sets of results | (their slots)
    double *const win = spmm_lds;
    double *const tvw = spmm_lds + (size_t)NV * W + (size_t)wave * tq;
    double *const yb = spmm_lds + (size_t)NV * W + (size_t)NW * tq;
    unsigned short *const tsw = reinterpret_cast<unsigned short *>(yb + 2 * NV * kGroupRows) + (size_t)wave * tq;
    // every XCD owns an eighth of the groups and its workgroups take them cyclically (k_spmm_pipe: neighbouring groups at the same
    // time, their windows overlap in the L2)
    const unsigned nwg = gridDim.x, per = (a.ngroups + 7u) / 8u, xcd = blockIdx.x % 8u;
    const unsigned gstride = nwg / 8u, gfirst = xcd * per + blockIdx.x / 8u;
    const unsigned gend = (xcd + 1u) * per < a.ngroups ? (xcd + 1u) * per : a.ngroups;
    const int npass = (a.nvec + NV - 1) / NV;
    if (a.b) {      // columns past the last vector
        for (unsigned g = gfirst; g < gend; g += gstride)
            if (tid < (unsigned)kSpmmCols && (int)tid >= a.nvec) a.partial[(size_t)g * kSpmmCols + tid] = 0.0;
    }
    if (gfirst >= gend) return;
    const unsigned short *const slots16 = reinterpret_cast<const unsigned short *>(a.sell.col16);

    if (tid < (unsigned)NV) win[(size_t)tid * W + (W - 1u)] = 0.0;      // the ZERO slot of every vector's window
    JpMeta M;
    JpAhead A;
    JpHead H;
    uint32_t c[(JJ + 1) / 2], cn[(JJ + 1) / 2];
    JpStage<JJ> T;
    // a group's rows: metadata, then (one wait) the heads and the first 256 entries behind them; whatever follows in further trips
    auto enter_group = [&](unsigned g) {
        jp_meta<OFFD>(a, g, A, M);
        const uint32_t pos_tail = jp_head_request(a, M, H);      // first entry behind the heads
        uint32_t tcnt = 0u;
        for (uint32_t e = K; e < M.len; ++e) tcnt += (uint32_t)__builtin_popcountll(__ballot(e < M.mylen));
        const uint32_t tail_lds = tcnt < tq ? tcnt : tq;      // (tcnt <= tq: launch_spmm_jpipe)
        double cv[4];
        unsigned cs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t idx = lane + 64u * (uint32_t)i;
            cv[i] = 0.0; cs[i] = 0u;
            if (idx < tail_lds && !(a.dbg & 16)) { cv[i] = a.sell.val[pos_tail + idx]; cs[i] = slots16[pos_tail + idx]; }
        }
        jp_head_finish(M, zero8, H);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t idx = lane + 64u * (uint32_t)i;
            if (idx < tail_lds) { tvw[idx] = cv[i]; tsw[idx] = (unsigned short)cs[i]; }
        }
        for (uint32_t idx = 256u + lane; idx < tail_lds; idx += 64u) { tvw[idx] = a.sell.val[pos_tail + idx]; tsw[idx] = slots16[pos_tail + idx]; }
    };
    jp_ahead(a, gfirst, tid, wave, A);
    jp_cols<JJ>(a, gfirst, tid, wruns, c);
#pragma unroll
    for (int h = 0; h < (JJ + 1) / 2; ++h) cn[h] = 0u;
    if (gfirst + gstride < gend) jp_cols<JJ>(a, gfirst + gstride, tid, wruns, cn);
    enter_group(gfirst);
    if (gfirst + gstride < gend) jp_ahead(a, gfirst + gstride, tid, wave, A);
    jp_stage_load<JJ>(a, gfirst, c, 0, T);
    jp_stage_store<JJ>(a, win, tid, T);
    __syncthreads();

    unsigned prev_g = 0xFFFFFFFFu, step = 0;      // the step whose results wait in LDS
    int prev_v0 = 0;
    auto flush_results = [&]() {                  // row g0 + tid of the previous step: consecutive lanes, consecutive addresses
        if (prev_g == 0xFFFFFFFFu || !a.ys || (a.dbg & 64)) return;
        const uint32_t r = prev_g * (uint32_t)kGroupRows + tid;
        const double *const src = yb + (size_t)((step + 1u) & 1u) * NV * kGroupRows;      // (written in step - 1)
        if (r < a.nrows) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (prev_v0 + v < a.nvec) *reinterpret_cast<double *>(reinterpret_cast<char *>(a.ys + (size_t)(prev_v0 + v) * a.vstride) + r * 8u) = src[v * kGroupRows + tid];
        }
    };
    for (unsigned g = gfirst; g < gend; g += gstride) {
        const bool more = g + gstride < gend;
        const unsigned rin = M.row - g * (unsigned)kGroupRows;        // the lane's row within the group
        for (int p = 0; p < npass; ++p) {
            const bool last = p + 1 == npass, have_next = !last || more;
            const int v0 = p * NV, nv = a.nvec - v0 < NV ? a.nvec - v0 : NV;
            // ---- the previous step's results (in front of this step's requests: the wait for those then covers no fresh store)
            flush_results();
            // ---- requests: the step's shifts and the rows' own x (used behind the products), then the next step's window
            double sg[NV], xself[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                sg[v] = (a.sigma && v < nv) ? a.sigma[v0 + v] : 0.0;
                xself[v] = (a.sigma && v < nv && M.live && !(a.dbg & 32)) ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(a.xs + (size_t)(v0 + v) * a.vstride) + M.row * 8u) : 0.0;
            }
            if (!(a.dbg & 1)) {
                if (!last) jp_stage_load<JJ>(a, g, c, v0 + NV, T);
                else if (more) jp_stage_load<JJ>(a, g + gstride, cn, 0, T);
            }
            // ---- this step: NV sums per lane out of the window, the head from registers, whatever follows from LDS
            double acc[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = 0.0;
#pragma unroll
            for (int e = 0; e < K; ++e) asm volatile("" : "+v"(H.s8[e]));     // (keeps the K x NV LDS addresses out of loop-invariant registers)
            const char *const cb = reinterpret_cast<const char *>(win);
            auto half = [&](int e0) {       // four entries' reads in flight, then their products in stored order
#pragma unroll
                for (int e4 = e0; e4 < e0 + K / 2; e4 += XR) {
                    double xr[XR][NV];
#pragma unroll
                    for (int i = 0; i < XR; ++i)
#pragma unroll
                        for (int v = 0; v < NV; ++v) xr[i][v] = *reinterpret_cast<const double *>(cb + (size_t)v * W * 8u + H.s8[e4 + i]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < XR; ++i)
#pragma unroll
                        for (int v = 0; v < NV; ++v) acc[v] = acc[v] + H.v[e4 + i] * xr[i][v];       // an absent entry adds 0.0 * 0.0
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (!(a.dbg & 2)) {
                half(0);
                if (M.len > (uint32_t)(K / 2)) half(K / 2);
            }
            uint32_t rel = 0u;                                        // entries of the slice behind the heads taken so far (wave-uniform)
            for (uint32_t k0 = K; k0 < M.len && !(a.dbg & 16); k0 += U) {
                double val[U];
                unsigned s8[U], idx[U];
                bool on[U];
#pragma unroll
                for (int e = 0; e < U; ++e) {
                    on[e] = k0 + e < M.mylen;
                    const unsigned long long m = __ballot(on[e]);
                    idx[e] = on[e] ? rel + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) : 0u;
                    rel += (uint32_t)__builtin_popcountll(m);
                }
#pragma unroll
                for (int e = 0; e < U; ++e) { val[e] = tvw[idx[e]]; s8[e] = tsw[idx[e]]; }      // (the launch made room for every slice's entries)
#pragma unroll
                for (int e = 0; e < U; ++e) { val[e] = on[e] ? val[e] : 0.0; s8[e] = on[e] ? s8[e] * 8u : zero8; }
#pragma unroll
                for (int e = 0; e < U; ++e)
#pragma unroll
                    for (int v = 0; v < NV; ++v)
                        acc[v] = acc[v] + val[e] * *reinterpret_cast<const double *>(cb + (size_t)v * W * 8u + s8[e]);
            }
            double r2[NV];
            double *const ydst = yb + (size_t)(step & 1u) * NV * kGroupRows;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                r2[v] = 0.0;
                double y = 0.0;
                if (v < nv && M.live) {
                    y = 0.0 + acc[v];                                 // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
                    if (OFFD) {
                        const double *xv = a.xs + (size_t)(v0 + v) * a.vstride;
                        double so = 0.0;
                        for (uint32_t k = M.oa; k < M.ob; ++k) so += a.offd.val[k] * xv[a.offd.col[k]];
                        y += so;                                      // second mult() call, src/matrix.c:440
                    }
                    if (a.sigma) y += sg[v] * xself[v];               // += sigma_j x_j (src/test_shifted.c:133)
                    if (a.b) { const double dd = (M.bi + (-1.0) * y) - 0.0; r2[v] = dd * dd; }
                }
                if (a.ys) ydst[v * kGroupRows + rin] = y;             // (the lanes' rows within the group are a permutation of 0 .. 255)
            }
            prev_g = g; prev_v0 = v0;
            ++step;
            if (a.b) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double t = wave_sum(r2[v]);
                    if (lane == 0) sm[wave * NV + v] = t;
                }
                __syncthreads();
                if ((int)tid < nv) {
                    double t = sm[tid];
                    for (int w = 1; w < NW; ++w) t += sm[w * NV + tid];
                    a.partial[(size_t)g * kSpmmCols + v0 + tid] = t;
                }
            }
            // ---- hand-over: nobody reads the window any more; the next step's has landed
            __syncthreads();
            if (have_next && !(a.dbg & 8)) jp_stage_store<JJ>(a, win, tid, T);
            __syncthreads();
        }
        if (more) {
#pragma unroll
            for (int h = 0; h < (JJ + 1) / 2; ++h) c[h] = cn[h];
            if (g + 2u * gstride < gend) jp_cols<JJ>(a, g + 2u * gstride, tid, wruns, cn);      // (requested in front of the heads: one wait)
            if (!(a.dbg & 4)) enter_group(g + gstride);               // (the heads are waited for right here: once per group)
            if (g + 2u * gstride < gend) jp_ahead(a, g + 2u * gstride, tid, wave, A);
        }
    }
    flush_results();
}

// Jagged slices with x windows (slots in col16), one 16-bit word of row / length per lane, a window of at most 2 046 slots whose
// columns lie within 16 bits of the group's first row.
hipError_t launch_spmm_jpipe(const SpmmArgs &a0, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    if (a0.ngroups == 0) return hipSuccess;
    if (!a0.sell.jag || !a0.sell.win_slots || !a0.sell.lane_info || !a0.sell.col16 || !a0.xs) return hipErrorInvalidValue;
    if (!a0.sell.win_list && !(a0.sell.win_runs && a0.sell.win_max_runs <= 64u)) return hipErrorInvalidValue;
    SpmmArgs a = a0;
    const unsigned W = (a0.sell.win_slots + 2u) & ~1u;                 // + the ZERO slot, which is the last one
    const int jj = (int)((W - 1u + 255u) / 256u);
    if (jj > 8) return hipErrorInvalidValue;
    a.wslots = W;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return hipErrorInvalidValue; }
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const unsigned ngroups_all = a0.ngroups;
    // LDS per workgroup: the window, two sets of results, and per wavefront `tq` entries (10 bytes each) of its slice behind the
    // rows' heads -- as many as the plan counted (SpmmArgs::tail_most) where three workgroups per CU leave room for them
    const unsigned fixed = (unsigned)kJpNV * W * 8u + 2u * (unsigned)kJpNV * (unsigned)kGroupRows * 8u, statics = 640u;
    const unsigned room = (160u * 1024u) / (unsigned)JPIPE_WAVES;
    unsigned tq = (a0.tail_most + 63u) & ~63u;
    if (fixed + statics + 40u * tq > room && fixed + statics + 40u * tq > 80u * 1024u) return hipErrorInvalidValue;      // (k_spmm_win takes such a block)
    const unsigned lds = fixed + 40u * tq;
    if (lds + statics > 160u * 1024u) return hipErrorInvalidValue;
    // one row of partial sums per group; the column sums run over spmm_grid(groups) rows: those beyond the last group are zero
    if (a.b) (void)hipMemsetAsync(a.partial + (size_t)ngroups_all * kSpmmCols, 0, sizeof(double) * 8 * kSpmmCols, st);
    auto go = [&](auto kernel) {
        static bool raised = false;
        static int occ = 0;
        static unsigned occ_lds = 0;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            (void)hipGetLastError();
            raised = true;
        }
        if (occ_lds != lds) {      // resident workgroups per CU
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, (size_t)lds) != hipSuccess) occ = 0;
            (void)hipGetLastError();
            occ_lds = lds;
        }
        int per_cu = occ;
        if (per_cu < 1) return hipErrorInvalidValue;
        if (per_cu > JPIPE_WAVES) per_cu = JPIPE_WAVES;
        if (const char *v = test_tok("spmm-jres")) per_cu = atoi(v) > 0 ? atoi(v) : per_cu;
        const unsigned resident = (unsigned)per_cu * (unsigned)cus;
        const unsigned grid = ngroups_all <= resident ? ((ngroups_all + 7u) & ~7u) : (resident & ~7u);
        if (e0 && e1) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, st, e0, e1, 0, a, tq);
        else hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, st, a, tq);
        return hipGetLastError();
    };
#define JP_GO(JJV) (with_offd ? go(k_spmm_jpipe<true, JJV>) : go(k_spmm_jpipe<false, JJV>))
    if (jj <= 6) return JP_GO(6);
    if (jj == 7) return JP_GO(7);
    return JP_GO(8);
#undef JP_GO
}

}  // namespace bicg
