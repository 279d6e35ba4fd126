// bicg_kernels.hip -- hand-written HIP kernels for the BiCGStab hot path on MI355X (gfx950, CDNA4).
//
// Everything here is HBM-bandwidth bound (0.15 flop/byte): no MFMA, the rules that matter are
// coalesced streaming of val/col, LDS staging, 64-wide wavefront reductions and few launches.
//
//   k_spmv        CSR "row-block stream" SpMV: a 256-thread workgroup owns a block of whole rows
//                 holding <= 2048 non-zeros, streams val/col fully coalesced (8 per thread, all
//                 loads in flight before the first use), gathers x, stages the products in LDS
//                 and reduces every row from LDS in stored order; fused dot-product epilogue.
//                 Replaces mult() + MPI_csr_spmv_ovlap (reference src/matrix.c:498-516, 428-441).
//   k_vec<...>    fused element-wise phases of the four iterations (replace the my_daxpy /
//                 my_dscal / my_dcopy / my_ddot call sequences of reference src/solver.c).
//   reductions    __shfl_down over the 64-lane wavefront -> LDS across the 4 wavefronts -> one
//                 partial per workgroup -> the LAST workgroup to arrive sums the partials in a
//                 fixed order (deterministic) and applies the scalar recurrence on the device.
//
// Compiled with -ffp-contract=off: every a*b+c keeps the two roundings of the reference's scalar
// loops, so the element-wise phases and every SpMV row are bit-identical to the CPU oracle; only
// the association of the dot-product sums differs.
#include "bicg_device.h"
#include "bicg_devfn.h"
#include "bicg_reduce.h"
#include "bicg_knobs.h"

#include <hip/hip_ext.h>   // hipExtLaunchKernelGGL: start/stop events bound to ONE kernel (roofline timing)
#include <cstdio>
#include <cstdlib>

// This file is compiled several times (Makefile: BICG_PART = 0..5, in parallel): the sliced-ELL launchers instantiate
// several hundred kernels and would otherwise serialise the build. Part 0 holds everything that is not a template
// (kernels and launch wrappers), parts 1-7 one group of sliced-ELL instantiations each; without BICG_PART the
// file is one translation unit.
#ifndef BICG_PART
#define BICG_PART -1
#endif
#define PART_IS(p) (BICG_PART == -1 || BICG_PART == (p))

namespace bicg {

// launch with optional per-kernel timing events (kernel-accurate, unlike events recorded around a launch)
// BICG_DEBUG=1: report a launch the runtime refused (or an error an earlier call left behind) where it happens
static inline void launch_debug(const char *what)
{
    static const bool debug = getenv("BICG_DEBUG") != nullptr;
    if (!debug) return;
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) fprintf(stderr, "bicgstab_hip: HIP error \"%s\" noticed at: %s\n", hipGetErrorString(err), what);
}
#define BICG_LAUNCH(kernel, ...)                 \
    do {                                         \
        hipLaunchKernelGGL(kernel, __VA_ARGS__); \
        launch_debug(#kernel);                   \
    } while (0)

template <class K, class... Args>
static void launch_timed_lds(K kernel, dim3 g, dim3 b, unsigned lds_bytes, hipStream_t st, hipEvent_t e0, hipEvent_t e1, Args... args)
{
    launch_debug("(left behind by an earlier call)");
    if (e0 && e1) hipExtLaunchKernelGGL(kernel, g, b, lds_bytes, st, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kernel, g, b, lds_bytes, st, args...);
    launch_debug(__PRETTY_FUNCTION__);
}
template <class K, class... Args>
static void launch_timed(K kernel, dim3 g, dim3 b, hipStream_t st, hipEvent_t e0, hipEvent_t e1, Args... args)
{
    launch_timed_lds(kernel, g, b, 0u, st, e0, e1, args...);
}

#if PART_IS(0)
__global__ void __launch_bounds__(kBlock) k_apply(Scal *S, int phase)
{
    if (S->done) return;
    apply_phase_block<true>(S, phase);
}
#endif

#if PART_IS(0)
__global__ void __launch_bounds__(kBlock) k_apply_p2p(Scal *S, int phase, int n, P2pRed pr, unsigned long long timeout_ticks)
{
    if (S->done) return;
    __shared__ double vals[kRedSlots * kMaxRanksP2p];
    __shared__ int s_fail;
    if (!p2p_collect(S, n, pr, timeout_ticks, vals, &s_fail)) return;
    if (phase != PH_NONE) apply_phase_block<true>(S, phase);
}
#endif

#if PART_IS(0)
__global__ void __launch_bounds__(kBlock) k_p2p_selftest(P2pRed pr, unsigned seq0, int rounds, unsigned long long timeout_ticks,
                                                         int *status)
{
    __shared__ double vals[kRedSlots * kMaxRanksP2p];
    __shared__ double expect[kRedSlots * kMaxRanksP2p];
    __shared__ int s_timeout;
    constexpr int n = kMaxDots;
    if (threadIdx.x == 0) s_timeout = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const unsigned seq = seq0 + (unsigned)r;
        for (int t = threadIdx.x; t < n * pr.nranks; t += kBlock) {
            const int p = t / n, d = t % n;
            ll_store(pr.mail[p] + mail_index(seq, pr.nranks, pr.rank, d), selftest_value(pr.rank, seq, d), seq);
        }
        for (int t = threadIdx.x; t < n * pr.nranks; t += kBlock) {
            const int p = t / n, d = t % n;
            double v;
            if (!ll_wait(pr.mail[pr.rank] + mail_index(seq, pr.nranks, p, d), seq, timeout_ticks, &v)) s_timeout = 1;
            vals[p * kRedSlots + d] = v;
            expect[p * kRedSlots + d] = selftest_value(p, seq, d);
        }
        __syncthreads();
        if (s_timeout) {                 // a peer is not answering: do not wait `rounds` time-outs
            if (threadIdx.x == 0) atomicAdd(&status[1], 1);
            return;
        }
        if ((int)threadIdx.x < n) {
            const double got = rank_tree_sum(vals + threadIdx.x, pr.nranks);
            const double want = rank_tree_sum(expect + threadIdx.x, pr.nranks);
            if (!(got == want)) atomicAdd(&status[0], 1);
        }
        __syncthreads();
    }
}
#endif

// stand-alone finisher (set-up phases, host reads, transports whose all-reduce the host enqueues)
#if PART_IS(0)
__global__ void __launch_bounds__(kBlock) k_finish(Scal *S, Finish f)
{
    __shared__ FinishLds L;
    (void)finish_group(S, f, f.roles, blockIdx.x, gridDim.x, L, nullptr);
}

void launch_finish(const Launch &L)
{
    BICG_LAUNCH(k_finish, dim3(L.fin.roles & FIN_SHARDS ? kShards : 1), dim3(kBlock), 0, L.st, L.S, L.fin);
}
#endif

// ------------------------------------------------------------------------------------------
// CSR SpMV, row-block stream
// ------------------------------------------------------------------------------------------
// CSR row-block stream kernel (ragged / long-row groups; the whole matrix when BICG_NO_SELL=1).
// Variants that were measured on Transport and dropped (profiles/r01_csr_baseline): an XCD-aware
// block mapping (fabric reads 386 -> 309 MB, wall time +3-5 %), 16-byte val/col loads (+2 us), a
// 2048-workgroup persistent grid-stride form (67.8 vs 60.5 us) and a software-pipelined persistent
// form that prefetches the next block's stream (67 us, flat in the number of resident workgroups):
// the kernel is bound by the vector L1's tag rate on the row-major x gather, which is what the
// sliced-ELL kernel below removes.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef double   f64x2 __attribute__((ext_vector_type(2)));

template <bool NT, class T> __device__ __forceinline__ T stream_load(const T *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

template <int NDOT, bool OFFD, bool NT, int MODE>
__global__ void __launch_bounds__(kBlock) k_spmv(SpmvArgs a)
{
    if (MODE == RED_WAVE) {
        __shared__ FinishLds fl;
        if (a.fin.seq && (blockIdx.x < (unsigned)kShards || (a.fin.roles & FIN_APPLY))) (void)finish_group(a.S, a.fin, a.fin.roles, blockIdx.x, gridDim.x, fl, nullptr);
    }
    // The sticky convergence flag is requested here but only consumed where state would be
    // modified (y stores, dot publication): an early `if (done) return` would put one more
    // dependent global load in front of every workgroup's stream.
    const int done = a.S->done;
    __shared__ __attribute__((aligned(16))) double prod[kChunk];
    __shared__ double sm[5 * (NDOT > 0 ? NDOT : 1)];

    const unsigned tid = threadIdx.x;
    const double *__restrict__ x = a.x;

    double acc[NDOT > 0 ? NDOT : 1];
#pragma unroll
    for (int d = 0; d < (NDOT > 0 ? NDOT : 1); ++d) acc[d] = 0.0;

    const unsigned first = blockIdx.x, last = a.nlist, step = gridDim.x;

    for (unsigned bi = first; bi < last; bi += step) {
        // one 16-byte descriptor per row block {first row, end row, first nnz, end nnz}: a single
        // wave-uniform load instead of the dependent chain rowblk -> ptr -> val/col
        const uint4 d = a.desc[bi];
        const uint32_t r0 = d.x, r1 = d.y, j0 = d.z, j1 = d.w;
        const uint32_t jw = j0;                                      // start of the staged window

        // this thread's (first) row: its pointers and dot operand are requested now, together with
        // the val/col stream, not after the barrier
        const uint32_t rme = r0 + tid;
        const bool mine = rme < r1;
        const uint32_t pa = mine ? a.diag.ptr[rme] : 0u, pb = mine ? a.diag.ptr[rme + 1] : 0u;
        uint32_t oa = 0u, ob = 0u;
        if (OFFD && mine) { oa = a.offd.ptr[rme]; ob = a.offd.ptr[rme + 1]; }
        double ume = 0.0;
        if (NDOT >= 1 && mine) ume = a.u[rme];

        auto finish_row = [&](uint32_t r, uint32_t a0, uint32_t a1, uint32_t o0, uint32_t o1, double ur) {
            double sum = 0.0;
            // 8 LDS reads in flight, then added in stored order (the order of reference
            // src/matrix.c:511-513); entries past the row end are replaced by +0.0, which leaves the
            // running sum unchanged bit for bit (a sum that is -0.0 would become +0.0, and the
            // reference's `0.0 + tempy` does that anyway)
            for (uint32_t k = a0; k < a1; k += 8) {
                double t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t kk = k + e < a1 ? k + e : a1 - 1;
                    t[e] = prod[kk];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += (k + e < a1) ? t[e] : 0.0;
            }
            double yi = 0.0 + sum;                           // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
            if (OFFD) {
                double so = 0.0;
                for (uint32_t k = o0; k < o1; ++k) so += a.offd.val[k] * x[a.offd.col[k]];
                yi += so;                                    // second mult() call, src/matrix.c:440
            }
            if (a.has_shift) yi += a.shift * x[r];           // (A + sigma I) x, src/shifted_solver.c:260
            if (!done) a.y[r] = yi;
            if (NDOT >= 1) acc[0] += ur * yi;
            if (NDOT == 2) acc[NDOT >= 2 ? 1 : 0] += yi * yi;
            if (NDOT == 3) acc[NDOT >= 2 ? 1 : 0] += ur * ur;
        };

        if (j1 - jw <= (uint32_t)kChunk) {
            // stream: all val/col loads of the block are issued before the first gather
            {
                uint32_t c[kNnzPerThread];
                double   v[kNnzPerThread];
#pragma unroll
                for (int i = 0; i < kNnzPerThread; ++i) {
                    const uint32_t j = j0 + tid + i * kBlock;
                    const bool ok = j < j1;
                    c[i] = ok ? stream_load<NT>(a.diag.col + j) : 0u;
                    v[i] = ok ? stream_load<NT>(a.diag.val + j) : 0.0;
                }
#pragma unroll
                for (int i = 0; i < kNnzPerThread; ++i) prod[tid + i * kBlock] = v[i] * x[c[i]];
            }
            __syncthreads();
            // one thread per row
            if (mine) finish_row(rme, pa - jw, pb - jw, oa, ob, ume);
            for (uint32_t r = rme + kBlock; r < r1; r += kBlock)     // blocks of very short rows
                finish_row(r, a.diag.ptr[r] - jw, a.diag.ptr[r + 1] - jw, OFFD ? a.offd.ptr[r] : 0u,
                           OFFD ? a.offd.ptr[r + 1] : 0u, NDOT >= 1 ? a.u[r] : 0.0);
        } else {
            // a single row longer than the chunk: strided partial sums + workgroup reduction
            double part[1] = {0.0};
            for (uint32_t j = j0 + tid; j < j1; j += kBlock) part[0] += a.diag.val[j] * x[a.diag.col[j]];
            block_sum<1>(part, sm);
            if (tid == 0) {
                double yi = 0.0 + part[0];
                if (OFFD) {
                    double so = 0.0;
                    for (uint32_t k = oa; k < ob; ++k) so += a.offd.val[k] * x[a.offd.col[k]];
                    yi += so;
                }
                if (a.has_shift) yi += a.shift * x[r0];
                if (!done) a.y[r0] = yi;
                if (NDOT >= 1) acc[0] += ume * yi;
                if (NDOT == 2) acc[NDOT >= 2 ? 1 : 0] += yi * yi;
                if (NDOT == 3) acc[NDOT >= 2 ? 1 : 0] += ume * ume;
            }
        }
        __syncthreads();   // prod is rewritten by the next row block
    }
    if (NDOT > 0 && !done) {
        if (MODE == RED_WAVE) wave_publish<(NDOT > 0 ? NDOT : 1)>(acc, a.red.partial, a.red.slot_base + blockIdx.x);
        else reduce_publish<(NDOT > 0 ? NDOT : 1), MODE == RED_TICKET_HEAVY>(acc, a.S, a.red, a.red.slot_base + blockIdx.x, sm);
    }
}

// One workgroup per row block: the hardware dispatcher balances the ~12k workgroups of a
// Transport-sized matrix better than a persistent grid; beyond kSpmvMaxGrid row blocks the kernel's
// loop strides.
#if PART_IS(0)
unsigned spmv_grid(uint32_t nlist)
{
    if (nlist == 0) return 0;
    return nlist < (uint32_t)kSpmvMaxGrid ? nlist : (unsigned)kSpmvMaxGrid;
}
#endif


#if PART_IS(0)
template <int NDOT, bool OFFD>
static void launch_spmv_var(const SpmvArgs &a, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    dim3 g(spmv_grid(a.nlist)), b(kBlock);
    const int mode = red_mode(a.red, a.fin, NDOT > 0);
    constexpr int HV = NDOT > 0 ? RED_TICKET_HEAVY : RED_TICKET;   // without dots there is no epilogue to be heavy
    if (mode == RED_WAVE) {
        if (a.nt) launch_timed(k_spmv<NDOT, OFFD, true, RED_WAVE>, g, b, st, e0, e1, a);
        else launch_timed(k_spmv<NDOT, OFFD, false, RED_WAVE>, g, b, st, e0, e1, a);
    } else if (mode == RED_TICKET_HEAVY) {
        if (a.nt) launch_timed(k_spmv<NDOT, OFFD, true, HV>, g, b, st, e0, e1, a);
        else launch_timed(k_spmv<NDOT, OFFD, false, HV>, g, b, st, e0, e1, a);
    } else {
        if (a.nt) launch_timed(k_spmv<NDOT, OFFD, true, RED_TICKET>, g, b, st, e0, e1, a);
        else launch_timed(k_spmv<NDOT, OFFD, false, RED_TICKET>, g, b, st, e0, e1, a);
    }
}

// ------------------------------------------------------------------------------------------
// Rows over lanes (long rows). Lane = row (sliced ELL) needs >= 256 rows per workgroup: a block of a few ten
// thousand rows with ~1000 entries each (synthetic banded CSR, half-bandwidth 512: 23 415 rows = 92 workgroups
// for 256 CUs, every lane walking 128 dependent batches) ran at 0.27 of the HBM roofline. Here a row is spread
// over T = 8..64 lanes (T from the block's mean row length, wave-uniform): lane l adds entries l, l + T, l + 2T, ...
// of its row in stored order, 8 loads of val and col in flight per lane, the T partial sums are combined by a
// fixed butterfly (DPP inside rows of 16 lanes, two bpermute steps above) -- so val / col are read fully
// coalesced straight from the CSR arrays (16-bit column offsets when they fit: 10 B per non-zero), a workgroup
// holds a few rows only (row blocks of <= 8192 non-zeros) and a 24 M-non-zero matrix makes ~3000 workgroups
// whatever its row length. The association of a row's sum differs from mult() (reference src/matrix.c:506-515):
// the result agrees to ~1e-16 x sum |a_ij x_j|, tested at 1e-13 (north_star: a stated tolerance); it is fixed, so
// runs are bit-reproducible. Taken when the block's rows average >= 128 entries and lane = row would leave the
// GPU short of workgroups (bicg_create; BICG_ROWSPLIT=0/1 overrides).
// ------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ double lanes_sum(double v)     // every lane of each aligned group of T lanes gets the group's sum
{
    v = dpp_add<0xB1, 0xF>(v);                   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);                   // quad_perm [2,3,0,1]
    if (T >= 8) v = dpp_add<0x141, 0xF>(v);      // row_half_mirror
    if (T >= 16) v = dpp_add<0x140, 0xF>(v);     // row_mirror
    if (T >= 32) v += __shfl_xor(v, 16);
    if (T >= 64) v += __shfl_xor(v, 32);
    return v;
}

template <int NDOT, bool OFFD, bool NT, bool C16, int T, int MODE>
__device__ __forceinline__ void rows_block(const SpmvArgs &a, uint32_t r0, uint32_t r1, int done, double (&acc)[NDOT > 0 ? NDOT : 1])
{
    constexpr int U = 8;
    constexpr uint32_t NSUB = kBlock / T;
    const unsigned tid = threadIdx.x, sub = tid / T, l = tid % T;
    const double *__restrict__ x = a.x;
    for (uint32_t rb = r0; rb < r1; rb += NSUB) {               // workgroup-uniform trip count
        const uint32_t r = rb + sub;
        const bool have = r < r1;
        const uint32_t pa = have ? a.diag.ptr[r] : 0u, pb = have ? a.diag.ptr[r + 1] : 0u;
        const uint32_t rs = have ? r : r0;                      // a safe x index for lanes without an entry
        double ur = 0.0;
        if (NDOT >= 1 && have && l == 0) ur = a.u[r];
        double s = 0.0;
        for (uint32_t j0 = pa + l; j0 < pb; j0 += U * T) {
            uint32_t c[U];
            double   v[U];
            // raw loads only inside the predicated part (a use of a loaded value there costs one round trip per
            // entry, see sell_row); out-of-range entries multiply 0.0 with x[row]
#pragma unroll
            for (int e = 0; e < U; ++e) {
                const uint32_t j = j0 + e * T;
                const bool ok = j < pb;
                v[e] = 0.0;
                if (C16) {
                    int d = 0;
                    if (ok) { d = NT ? __builtin_nontemporal_load(a.diag_col16 + j) : a.diag_col16[j]; v[e] = stream_load<NT>(a.diag.val + j); }
                    c[e] = rs + (uint32_t)d;
                } else {
                    c[e] = rs;
                    if (ok) { c[e] = stream_load<NT>(a.diag.col + j); v[e] = stream_load<NT>(a.diag.val + j); }
                }
            }
            double xv[U];
#pragma unroll
            for (int e = 0; e < U; ++e) xv[e] = x[c[e]];
#pragma unroll
            for (int e = 0; e < U; ++e) s += v[e] * xv[e];
        }
        s = lanes_sum<T>(s);
        if (have && l == 0) {
            double yi = 0.0 + s;                                 // y = 0 ; y += tempy  (src/matrix.c:434-437)
            if (OFFD) {
                double so = 0.0;
                for (uint32_t k = a.offd.ptr[r]; k < a.offd.ptr[r + 1]; ++k) so += a.offd.val[k] * x[a.offd.col[k]];
                yi += so;                                        // second mult() call, src/matrix.c:440
            }
            if (a.has_shift) yi += a.shift * x[r];
            if (!done) a.y[r] = yi;
            if (NDOT >= 1) acc[0] += ur * yi;
            if (NDOT == 2) acc[NDOT >= 2 ? 1 : 0] += yi * yi;
            if (NDOT == 3) acc[NDOT >= 2 ? 1 : 0] += ur * ur;
        }
    }
}

template <int NDOT, bool OFFD, bool NT, bool C16, int MODE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MODE == RED_TICKET ? 8 : 4, 8))) k_spmv_rows(SpmvArgs a)
{
    if (MODE == RED_WAVE) {
        __shared__ FinishLds fl;
        if (a.fin.seq && (blockIdx.x < (unsigned)kShards || (a.fin.roles & FIN_APPLY))) (void)finish_group(a.S, a.fin, a.fin.roles, blockIdx.x, gridDim.x, fl, nullptr);
    }
    const int done = a.S->done;
    __shared__ double sm[5 * (NDOT > 0 ? NDOT : 1)];
    double acc[NDOT > 0 ? NDOT : 1];
#pragma unroll
    for (int d = 0; d < (NDOT > 0 ? NDOT : 1); ++d) acc[d] = 0.0;
    for (unsigned bi = blockIdx.x; bi < a.nlist; bi += gridDim.x) {
        const uint4 d = a.desc[bi];
        const uint32_t nr = d.y - d.x, mean = nr ? (d.w - d.z) / nr : 0u;
        // lanes per row from the block's mean row length (workgroup-uniform): ~8 entries per lane and batch
        if (mean >= 320u) rows_block<NDOT, OFFD, NT, C16, 64, MODE>(a, d.x, d.y, done, acc);
        else if (mean >= 160u) rows_block<NDOT, OFFD, NT, C16, 32, MODE>(a, d.x, d.y, done, acc);
        else if (mean >= 80u) rows_block<NDOT, OFFD, NT, C16, 16, MODE>(a, d.x, d.y, done, acc);
        else rows_block<NDOT, OFFD, NT, C16, 8, MODE>(a, d.x, d.y, done, acc);
    }
    if (NDOT > 0 && !done) {
        if (MODE == RED_WAVE) wave_publish<(NDOT > 0 ? NDOT : 1)>(acc, a.red.partial, a.red.slot_base + blockIdx.x);
        else reduce_publish<(NDOT > 0 ? NDOT : 1), MODE == RED_TICKET_HEAVY>(acc, a.S, a.red, a.red.slot_base + blockIdx.x, sm);
    }
}

template <int NDOT, bool OFFD>
static void launch_spmv_rows_var(const SpmvArgs &a, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    dim3 g(spmv_grid(a.nlist)), b(kBlock);
    const int mode = red_mode(a.red, a.fin, NDOT > 0);
    constexpr int HV = NDOT > 0 ? RED_TICKET_HEAVY : RED_TICKET;
    const bool c16 = a.diag_col16 != nullptr, nt = a.nt != 0;
#define ROWS_GO(MD)                                                                                        \
    do {                                                                                                   \
        if (c16) { if (nt) launch_timed(k_spmv_rows<NDOT, OFFD, true, true, MD>, g, b, st, e0, e1, a);       \
                   else launch_timed(k_spmv_rows<NDOT, OFFD, false, true, MD>, g, b, st, e0, e1, a); }      \
        else { if (nt) launch_timed(k_spmv_rows<NDOT, OFFD, true, false, MD>, g, b, st, e0, e1, a);          \
               else launch_timed(k_spmv_rows<NDOT, OFFD, false, false, MD>, g, b, st, e0, e1, a); }         \
    } while (0)
    if (mode == RED_WAVE) ROWS_GO(RED_WAVE);
    else if (mode == RED_TICKET_HEAVY) ROWS_GO(HV);
    else ROWS_GO(RED_TICKET);
#undef ROWS_GO
}

static bool launch_spmv_rows(const SpmvArgs &a, int ndot, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    if (with_offd) {
        if (ndot == 0) launch_spmv_rows_var<0, true>(a, st, e0, e1); else if (ndot == 1) launch_spmv_rows_var<1, true>(a, st, e0, e1);
        else if (ndot == 2) launch_spmv_rows_var<2, true>(a, st, e0, e1); else launch_spmv_rows_var<3, true>(a, st, e0, e1);
    } else {
        if (ndot == 0) launch_spmv_rows_var<0, false>(a, st, e0, e1); else if (ndot == 1) launch_spmv_rows_var<1, false>(a, st, e0, e1);
        else if (ndot == 2) launch_spmv_rows_var<2, false>(a, st, e0, e1); else launch_spmv_rows_var<3, false>(a, st, e0, e1);
    }
    return true;
}

unsigned g_product_kernels = 0;

bool launch_spmv(const SpmvArgs &a, int ndot, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    if (a.nlist == 0) return false;
    g_product_kernels |= a.rowsplit ? PK_ROWS : PK_CSR;
    if (a.rowsplit) return launch_spmv_rows(a, ndot, with_offd, st, e0, e1);
    if (with_offd) {
        if (ndot == 0) launch_spmv_var<0, true>(a, st, e0, e1); else if (ndot == 1) launch_spmv_var<1, true>(a, st, e0, e1);
        else if (ndot == 2) launch_spmv_var<2, true>(a, st, e0, e1); else launch_spmv_var<3, true>(a, st, e0, e1);
    } else {
        if (ndot == 0) launch_spmv_var<0, false>(a, st, e0, e1); else if (ndot == 1) launch_spmv_var<1, false>(a, st, e0, e1);
        else if (ndot == 2) launch_spmv_var<2, false>(a, st, e0, e1); else launch_spmv_var<3, false>(a, st, e0, e1);
    }
    return true;
}
#endif

// ------------------------------------------------------------------------------------------
// Sliced-ELL SpMV (the default path for rows whose slice pads by < 25 %)
//
// Why: the CSR row-block kernel above is limited by the vector L1 (TCP), not by HBM: with lanes
// walking the non-zeros row-major, the x gather of a wavefront touches ~20 different cache lines
// per instruction; rocprofv3 shows 26 M TCP tag accesses per SpMV (1.1 per non-zero), the TCP
// clock-enabled 85 % of the kernel, and the time does not react to fabric traffic or occupancy.
// With lane = row (SELL-64) consecutive lanes read consecutive entries of val/col AND, for banded
// matrices, consecutive entries of x: ~0.35 tag accesses per non-zero, no LDS, no barrier.
// Each lane accumulates its own row in stored order -> bit-identical to mult() (reference
// src/matrix.c:506-515) for every row. Padding entries are loaded (coalescing) but never added.
// ------------------------------------------------------------------------------------------
typedef short i16x4 __attribute__((ext_vector_type(4)));

// C16: column indices are read as 16-bit offsets from the row (col = row + delta), four
// consecutive entries of a lane packed in one 8-byte word: 10 instead of 12 bytes per non-zero.
// Used when every entry of the sliced-ELL copy satisfies |col - row| < 32768 (banded matrices).
// y_i of this lane's row of list entry gi (diag part in stored order, then the offd part, then the shift):
// the body shared by the SpMV kernel and the SpMV-with-epilogue kernel below
extern __shared__ double dyn_lds[];      // the x window of LAY_JAGW launches (SellDev::win_slots doubles)

// LAY_JAGW: copy the x values the group's rows touch into LDS (all 256 threads; the caller's loop is workgroup-uniform)
__device__ __forceinline__ void sell_stage_window(const SpmvArgs &a, unsigned g, double *win)
{
    __syncthreads();                                          // the previous group's reads of the window are done
    const uint32_t r0 = a.sell.win_ptr[g], r1 = a.sell.win_ptr[g + 1];
    for (uint32_t r = r0; r < r1; ++r) {
        const uint2 run = a.sell.win_runs[r];                 // wave-uniform
        const uint32_t len = run.y & 0xFFFFu, slot0 = run.y >> 16;
        for (uint32_t i = threadIdx.x; i < len; i += kBlock) win[slot0 + i] = a.x[run.x + i];
    }
    __syncthreads();
}

// A list-driven slice whose descriptor is known (SellDev::sdesc): the sum of one row of a CONSTANT slice (MASKED = false: every
// row has every entry of the list) or of a MASKED slice (pm = the entries this lane's row has). Distances and values arrive as
// scalar loads of whole batches (the lists are padded with zeros to a multiple of 8 + 16), x is addressed as
// (uniform base) + (32-bit byte offset of the row): two vector instructions of arithmetic per entry next to its load,
// where the general loop spent twenty-odd on a slice of the 7-point Laplacian. The entries are added in list = stored order, the
// ones a row does not have are not added: the same sum, bit for bit, as the general loop's (reference src/matrix.c:506-515).
// (the lists, the descriptors and the group list are read-only for every launch: loads through the constant address space are
// scalar loads whatever the compiler can prove about the stores of the kernel)
#define BICG_KCONST __attribute__((address_space(4)))
struct SellPre { unsigned g; uint4 d; uint32_t pm; };      // group, descriptor and row mask of the wavefront's slice, requested ahead by the product
typedef int sell_i8 __attribute__((ext_vector_type(8)));
typedef unsigned sell_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 sell_desc_load(const uint4 *p)       // (a native vector type: uint4's copy constructor would drop the address space)
{
    const sell_u4 v = *(const BICG_KCONST sell_u4 *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool MASKED>
__device__ __forceinline__ double sell_list_row(const double *__restrict__ x, uint32_t row, uint32_t len, const int *uo_g, const double *uv_g,
                                                uint32_t pm)
{
    const BICG_KCONST int *uo = (const BICG_KCONST int *)uo_g;
    const BICG_KCONST double *uv = (const BICG_KCONST double *)uv_g;
    const uint32_t boff = row << 3;                               // rows < 2^29 (build_slice_desc)
    double sum = 0.0;
    for (uint32_t k0 = 0; k0 < len; k0 += 8) {
        const sell_i8 o = *(const BICG_KCONST sell_i8 *)(uo + k0);
        double v[8], xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = uv[k0 + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (MASKED) {
                // an absent neighbour may lie outside the vector: the lane reads x of its own row instead
                const uint32_t off = ((pm >> (k0 + e)) & 1u) ? (uint32_t)o[e] << 3 : 0u;
                xv[e] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + (uint32_t)(boff + off));
            } else {
                xv[e] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x + o[e]) + boff);   // (padding: distance 0)
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (MASKED ? ((pm >> (k0 + e)) & 1u) != 0u : k0 + e < len) sum += v[e] * xv[e];
    }
    return sum;
}

// The same sum for a list of exactly N <= 8 entries (SellDev::all_lists): no loop, no tests of the length; the distances arrive as
// byte offsets (SellDev::uoff8) and are added to the row's byte offset modulo 2^32 -- one vector addition per entry, the load
// takes (uniform base of x) + (32-bit offset). o8 / v: the list, already in scalar registers (the caller keeps the list of the
// previous slice: the interior of a stencil has ONE).
template <int N, bool MASKED>
__device__ __forceinline__ double sell_list_fixed(const double *x, uint32_t boff, const sell_i8 &o8, const double (&v)[8], uint32_t pm,
                                                  double *yp, double yv)
{
    double xv[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const uint32_t off = (!MASKED || ((pm >> e) & 1u)) ? (uint32_t)o8[e] : 0u;      // (an absent neighbour may lie outside the vector)
        xv[e] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + (uint32_t)(boff + off));
    }
    // the PREVIOUS slice's result is stored here, behind this slice's gathers: issued right after its own slice it would be the
    // youngest request in flight when the loop comes round, and the wait for the next descriptor (vector loads return in order)
    // a wait for the store's acknowledgement. (Unconditional: with a path that does not store, the wait for the last gather
    // becomes a wait for everything.)
    *yp = yv;
    double sum = 0.0;
#pragma unroll
    for (int e = 0; e < N; ++e)
        if (!MASKED || ((pm >> e) & 1u) != 0u) sum += v[e] * xv[e];                       // list = stored order
    return sum;
}
template <bool MASKED>
__device__ __forceinline__ double sell_list_switch(uint32_t len, const double *x, uint32_t boff, const sell_i8 &o8, const double (&v)[8], uint32_t pm,
                                                   double *yp, double yv)
{
    switch (len) {                                                // wave-uniform
    case 1: return sell_list_fixed<1, MASKED>(x, boff, o8, v, pm, yp, yv);
    case 2: return sell_list_fixed<2, MASKED>(x, boff, o8, v, pm, yp, yv);
    case 3: return sell_list_fixed<3, MASKED>(x, boff, o8, v, pm, yp, yv);
    case 4: return sell_list_fixed<4, MASKED>(x, boff, o8, v, pm, yp, yv);
    case 5: return sell_list_fixed<5, MASKED>(x, boff, o8, v, pm, yp, yv);
    case 6: return sell_list_fixed<6, MASKED>(x, boff, o8, v, pm, yp, yv);
    case 7: return sell_list_fixed<7, MASKED>(x, boff, o8, v, pm, yp, yv);
    default: return sell_list_fixed<8, MASKED>(x, boff, o8, v, pm, yp, yv);
    }
}

template <bool OFFD, bool NT, int LAY, bool LL, int U = 8>      // U entries per lane in flight (4, 8, 16 measured identical on Transport)
__device__ __forceinline__ double sell_row(const SpmvArgs &a, unsigned gi, int done, uint32_t &row, bool &live, bool &ll_failed,
                                           const double *win = nullptr, bool have_pre = false, SellPre pre = SellPre{0u, {0u, 0u, 0u, 0u}, 0u})
{
    constexpr bool WIN = LAY == LAY_JAGW, C16 = (LAY & 1) != 0 || WIN, JAG = LAY >= LAY_JAG32 && LAY <= LAY_JAGW, CONSTV = LAY >= LAY_PAD32C;
    // (the wavefront's number as a SCALAR: the slice's base, length and list positions then come through the scalar cache and the
    // tests on them are scalar branches -- as a vector value the compiler masked and unmasked lanes around every entry)
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const double *__restrict__ x = a.x;
    const unsigned g = have_pre ? pre.g : (a.glist ? a.glist[gi] : gi);
    row = g * kGroupRows + tid;                                // = slice * 64 + lane
    if (LAY == LAY_JAGW && a.sell.perm) row = g * kGroupRows + a.sell.perm[(size_t)g * kGroupRows + tid];
    const uint32_t slice = g * (kGroupRows / kSliceRows) + wave;
    live = row < a.nrows;

    // constant and masked slices by their descriptor (SellDev::sdesc): all 64 rows exist, nothing else of the slice's
    // metadata is read
    bool listed = false;
    double lsum = 0.0;
    if (CONSTV && a.sell.sdesc && slice * kSliceRows < a.nrows) {
        uint4 d;
        if (have_pre) d = pre.d;
        else {
            d = sell_desc_load(a.sell.sdesc + slice);
        }
        const uint32_t kind = d.x >> 16, dlen = d.x & 0xFFFFu;
        if (kind == kSliceConstant) {
            listed = true;
            lsum = sell_list_row<false>(x, row, dlen, a.sell.uoff + d.y, a.sell.uval + d.z, 0u);
        } else if (kind == kSliceMasked) {
            listed = true;
            const uint32_t pm = have_pre ? pre.pm : (uint32_t)a.sell.rmask[(size_t)d.w * kSliceRows + lane];
            lsum = sell_list_row<true>(x, row, dlen, a.sell.uoff + d.y, a.sell.uval + d.z, pm);
        }
    }

    uint32_t base = 0u, len = 0u, base16 = 0u;
    if (slice * kSliceRows < a.nrows && !(CONSTV && listed)) {
        base = a.sell.slice_base[slice]; len = a.sell.slice_len[slice];
        if (C16 && !JAG) base16 = a.sell.slice_base16[slice];
    }
    // uniform slice: the columns are row + uoff[k], one list for the whole slice (scalar loads) -- no col / col16 traffic.
    // Same loop as the padded slices (one body: a second copy of it cost the ticket-mode kernels three spilled registers)
    const int *__restrict__ uo = nullptr;                         // padded with zeros to a multiple of U (+ U)
    const double *__restrict__ uv = nullptr;                      // constant slice: the values too (padded with zeros alike)
    if (!JAG && a.sell.ubase && slice * kSliceRows < a.nrows && !(CONSTV && listed)) {
        const uint32_t ub = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.sell.ubase[slice]);
        if (ub != 0xFFFFFFFFu) {
            uo = a.sell.uoff + ub;
            if (CONSTV) {
                const uint32_t vb = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.sell.vbase[slice]);
                if (vb != 0xFFFFFFFFu) uv = a.sell.uval + vb;
            }
        }
    }
    // masked slice (SellDev::mbase): list of (distance, value) pairs + one word per row saying which of them the row has
    bool masked = false;
    uint32_t pm = 0u;
    if (CONSTV && uv && a.sell.mbase) {
        const uint32_t mb = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.sell.mbase[slice]);
        if (mb != 0xFFFFFFFFu) {
            masked = true;
            len = mb >> 26;                                       // the list's length, not the longest row's
            pm = a.sell.rmask[(size_t)(mb & 0x03FFFFFFu) * kSliceRows + lane];
        }
    }
    // (a list-driven slice knows its rows' lengths: all equal to the slice's, or given by the masks -- no row-pointer loads)
    uint32_t mylen = 0u, oa = 0u, ob = 0u;
    if (live) {
        mylen = (uo || (CONSTV && listed)) ? len : a.diag.ptr[row + 1] - a.diag.ptr[row];
        if (OFFD && (!LL || gi >= a.ll.first_bnd)) { oa = a.offd.ptr[row]; ob = a.offd.ptr[row + 1]; }
    }

    double sum = 0.0;
    if (JAG) {
        // Jagged slice: step k of the slice holds the entries of the lanes whose row has more than k entries, and
        // only those, in lane order -- no padding is stored or read. A lane's entry is at (entries of the earlier
        // steps) + (live lanes below it): a ballot, a population count and mbcnt, all from the row lengths, so
        // every load of a batch is still issued back to back. With equal row lengths this IS the padded layout.
        const uint32_t rb = live ? row : 0u;
        uint32_t pos = base;                                  // wave-uniform: first entry of step k
        for (uint32_t k0 = 0; k0 < len; k0 += U) {
            uint32_t c[U];
            double   v[U];
            bool     mine[U];
            // The val / col loads are predicated per lane, and NOTHING that depends on a loaded value sits inside
            // the predicated block (the compiler waits for a load before the block ends otherwise: one round trip per
            // entry instead of one per batch). The gathers are unpredicated for the same reason: a lane whose row
            // has ended reads x of its own row.
#pragma unroll
            for (int e = 0; e < U; ++e) {
                mine[e] = k0 + e < mylen;
                const unsigned long long m = __ballot(mine[e]);
                const uint32_t j = pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                pos += (uint32_t)__builtin_popcountll(m);
                c[e] = 0u; v[e] = 0.0;
                if (mine[e]) {
                    if (WIN) {
                        const unsigned short *sl = reinterpret_cast<const unsigned short *>(a.sell.col16);
                        c[e] = NT ? __builtin_nontemporal_load(sl + j) : sl[j];
                    } else if (C16) {
                        c[e] = (uint32_t)(int)(NT ? __builtin_nontemporal_load(a.sell.col16 + j) : a.sell.col16[j]);
                    } else {
                        c[e] = NT ? __builtin_nontemporal_load(a.sell.col + j) : a.sell.col[j];
                    }
                    v[e] = NT ? __builtin_nontemporal_load(a.sell.val + j) : a.sell.val[j];
                }
            }
            double xv[U];
#pragma unroll
            for (int e = 0; e < U; ++e) {
                if (WIN) xv[e] = win[c[e]];                   // slot 0 for a lane whose row has ended
                else xv[e] = x[mine[e] ? (C16 ? rb + c[e] : c[e]) : rb];
            }
#pragma unroll
            for (int e = 0; e < U; ++e)
                if (mine[e]) sum += v[e] * xv[e];             // stored order
        }
    }
    for (uint32_t k0 = 0; !JAG && k0 < len; k0 += U) {
        uint32_t c[U];
        double   v[U];
        // lanes past the last row hold padding only: their offsets are 0 and must not turn into
        // reads of x[row >= nrows] (the vector may end before the 64-row slice does)
        const uint32_t rb = live ? row : 0u;
        if (CONSTV && masked) {
#pragma unroll
            for (int e = 0; e < U; ++e) c[e] = rb + (((pm >> (k0 + e)) & 1u) ? (uint32_t)uo[k0 + e] : 0u);   // an absent neighbour may lie outside the vector
        } else if (uo) {
#pragma unroll
            for (int e = 0; e < U; ++e) c[e] = rb + (uint32_t)uo[k0 + e];
        } else if (C16) {
            static_assert(U % 4 == 0, "packed 16-bit columns come four at a time");
#pragma unroll
            for (int q = 0; q < U / 4; ++q) {
                const bool ok = k0 + 4 * q < len;             // wave-uniform; the quad is padded
                const i16x4 *p = reinterpret_cast<const i16x4 *>(a.sell.col16) +
                                 ((size_t)base16 / 4 + (size_t)((k0 / 4) + q) * kSliceRows + lane);
                i16x4 dq = (i16x4)(0);
                if (ok) dq = NT ? __builtin_nontemporal_load(p) : *p;
                c[4 * q + 0] = rb + (int)dq.x; c[4 * q + 1] = rb + (int)dq.y;
                c[4 * q + 2] = rb + (int)dq.z; c[4 * q + 3] = rb + (int)dq.w;
            }
        }
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const bool ok = k0 + e < len;                     // wave-uniform
            const uint32_t j = base + (k0 + e) * kSliceRows + lane;
            if (!C16 && !uo) c[e] = ok ? (NT ? __builtin_nontemporal_load(a.sell.col + j) : a.sell.col[j]) : 0u;
            if (CONSTV && uv) v[e] = uv[k0 + e];              // (wave-uniform branch, scalar load)
            else v[e] = ok ? (NT ? __builtin_nontemporal_load(a.sell.val + j) : a.sell.val[j]) : 0.0;
        }
        double xv[U];
#pragma unroll
        for (int e = 0; e < U; ++e) xv[e] = x[c[e]];
#pragma unroll
        for (int e = 0; e < U; ++e)
            if ((CONSTV && masked) ? ((pm >> (k0 + e)) & 1u) != 0u : k0 + e < mylen) sum += v[e] * xv[e];   // stored order; padding never added
    }
    if (CONSTV && listed) sum = lsum;
    double yi = 0.0 + sum;                                    // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
    if (OFFD) {
        double so = 0.0;
        for (uint32_t k = oa; k < ob; ++k) {
            double xh;
            if (LL) {
                // the value comes straight from the landing ring; the diag part above ran while it travelled
                xh = 0.0;
                if (!done && !ll_wait(a.ll.ring + ((size_t)(a.ll.seq % kHaloRing) * a.ll.halo + (a.offd.col[k] - a.nrows)) * 2,
                                      a.ll.seq, a.ll.timeout_ticks, &xh))
                    ll_failed = true;
            } else {
                xh = x[a.offd.col[k]];
            }
            so += a.offd.val[k] * xh;
        }
        yi += so;                                             // second mult() call, src/matrix.c:440
    }
    if (a.has_shift && live) yi += a.shift * x[row];          // (A + sigma I) x, src/shifted_solver.c:260
    return yi;
}

// the leading workgroups of a launch with in-kernel halo exchange: store the send list into the peers' landing rings
__device__ __forceinline__ void sell_halo_push(const SpmvArgs &a, unsigned bid, int done)
{
    if (done) return;
    for (uint32_t i = bid * kBlock + threadIdx.x; i < a.ll.nsend; i += a.ll.npush * kBlock)
        ll_store(reinterpret_cast<llword *>(a.ll.dst0[i] + (unsigned long long)(a.ll.seq % kHaloRing) * a.ll.dstride[i]),
                 a.x[a.ll.send_idx[i]], a.ll.seq);
}

template <int NDOT, bool OFFD, bool NT, int LAY, bool LL, int MODE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((MODE == RED_TICKET && LAY < LAY_PAD32C) ? 8 : 4, 8))) k_spmv_sell(SpmvArgs a)   // (layouts with list-driven slices: no pin, they spilled 200 bytes per lane at 64 registers)
{
    const int done = a.S->done;       // consumed at the stores only (see k_spmv)
    __shared__ double sm[5 * (NDOT > 0 ? NDOT : 1)];
    unsigned bid = blockIdx.x, nblocks = gridDim.x;
    bool ll_failed = false;
    if (LL) {
        // peer-to-peer exchange inside the launch: the leading workgroups are scheduled first and
        // send; x is complete (it was written by earlier kernels), so nothing has to be waited for
        if (bid < a.ll.npush) { sell_halo_push(a, bid, done); return; }
        bid -= a.ll.npush; nblocks -= a.ll.npush;
    }
    if (MODE == RED_WAVE) {
        // a dot group of EARLIER kernels rides on this launch: its first workgroups add up the shards
        // (and hand the sums to the other ranks) while everybody else already streams the matrix
        // (only the workgroups that have a part in it: the others would still wait for `done` inside finish_group before their
        // first row -- one round trip per workgroup, 40.4 instead of 33.1 us per product of the pipelined iteration)
        __shared__ FinishLds fl;
        if (a.fin.seq && (bid < (unsigned)kShards || (a.fin.roles & FIN_APPLY))) (void)finish_group(a.S, a.fin, a.fin.roles, bid, nblocks, fl, nullptr);
    }

    double acc[NDOT > 0 ? NDOT : 1];
#pragma unroll
    for (int d = 0; d < (NDOT > 0 ? NDOT : 1); ++d) acc[d] = 0.0;

    // Which groups: workgroup bid stands for VIRTUAL workgroup vb of the canonical order -- XCD-contiguous (workgroup b runs on
    // XCD b % 8: XCD x gets the x-th eighth of the virtual workgroups, so one L2 fetches what neighbouring groups share) and,
    // every other product, reversed -- and takes the CONTIGUOUS groups vb * each ... of the list. Its partial sums go to slot vb
    // whichever physical workgroup computed them: the association of a dot sum does not depend on placement. Direction: with ONE
    // group per workgroup it does not matter either; with several (each > 1, grids beyond 65 536 groups) a reversed launch adds a
    // workgroup's groups in the opposite order -- a different, equally fixed association, and every solve / stand-alone call starts
    // from the same direction (bicg_ctx::spmv_dir is reset there), so repeated calls on one context give the same bits.
    // (Launches with the halo exchange inside keep the strided assignment: their leading workgroups are the senders.)
    unsigned vb = bid;
    if (a.xcd_map && !LL && bid < (nblocks / 8u) * 8u) vb = (bid % 8u) * (nblocks / 8u) + bid / 8u;
    if (a.reverse && !LL) vb = nblocks - 1u - vb;
    const unsigned each = LL ? 1u : (a.nlist + nblocks - 1u) / nblocks;
    const unsigned slot = LL ? bid : vb;
    const unsigned gfirst = LL ? bid : vb * each, gend = LL ? a.nlist : (gfirst + each < a.nlist ? gfirst + each : a.nlist);
    // Blocks with list-driven slices (SellDev::sdesc): what a wavefront needs to know about its next slices is requested while
    // it multiplies the current one -- the group number three groups ahead, the descriptor two ahead, the row masks one ahead --
    // so that a constant or masked slice is the scalar loads of its lists (scalar cache) and ONE vector round trip, the x gathers,
    // instead of four dependent trips (group, metadata, lists / masks, x). The requests are VECTOR loads of wave-uniform
    // addresses on purpose: vector loads return in order and are waited for by count, so they stay in flight across the product
    // of the current slice; scalar loads can only be waited for all at once, i.e. at the very next scalar load.
    constexpr bool PRE = LAY >= LAY_PAD32C && !LL;
    const bool pre_on = PRE && a.sell.sdesc != nullptr;
    unsigned vzero = 0u;
    if (PRE) asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));      // a zero the compiler takes for lane-dependent
    const unsigned wvec = threadIdx.x >> 6;
    // Every slice list-driven, lists of at most 8 entries (SellDev::all_lists -- a constant-coefficient stencil): a loop of its own
    // below. Its four wavefronts do not take four CONSECUTIVE slices but four slices `ystride` apart (SellDev::ystride = slices
    // per grid line: the same x segment of four consecutive grid lines) -- the +line gather of one wavefront is then the own-row
    // gather of the next, through the CU's L1 while both are in flight (56 instead of 80 cache lines per four slices of the
    // 7-point Laplacian). Any ystride gives each slice to exactly one wavefront: slice = (g / S) 4 S + wave S + g % S.
    const bool lean = PRE && pre_on && a.sell.all_lists != 0 && !OFFD && !a.has_shift;
    const unsigned ys = lean ? (unsigned)a.sell.ystride : 0u;
    auto slice_of = [&](unsigned g, unsigned w) -> uint32_t {
        const unsigned sh = (unsigned)__builtin_ctz(ys | 0x80000000u);                         // ystride is a power of two (or 0)
        return ys ? ((g >> sh) << (sh + 2u)) + (w << sh) + (g & (ys - 1u)) : g * (kGroupRows / kSliceRows) + w;
    };
    auto group_idx = [&](unsigned q) -> unsigned { return a.reverse ? gfirst + (gend - 1u - q) : q; };      // (q < gend)
    auto group_vec = [&](unsigned q) -> unsigned { return a.glist ? a.glist[group_idx(q) + vzero] : group_idx(q); };
    auto desc_vec = [&](unsigned g) -> uint4 {
        const uint32_t sl = slice_of(g, wvec);
        uint4 d = make_uint4(0u, 0u, 0u, 0u);
        if (sl * kSliceRows < a.nrows) d = a.sell.sdesc[sl];
        return d;
    };
    auto first_lane = [](uint4 v) -> uint4 {
        return make_uint4((uint32_t)__builtin_amdgcn_readfirstlane((int)v.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y),
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)v.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)v.w));
    };
    auto mask_vec = [&](const uint4 &d) -> uint32_t {
        return (d.x >> 16) == (uint32_t)kSliceMasked ? (uint32_t)a.sell.rmask[(size_t)d.w * kSliceRows + (threadIdx.x & 63u)] : 0u;
    };
    SellPre pcur = {0u, make_uint4(0u, 0u, 0u, 0u), 0u};
    unsigned g1 = 0u, gv2 = 0u;                                   // group of the next slice (scalar), of the one after it (as loaded)
    uint4 dv1 = make_uint4(0u, 0u, 0u, 0u);                       // descriptor of the next slice (as loaded)
    if (PRE && pre_on && gfirst < gend) {
        pcur.g = (unsigned)__builtin_amdgcn_readfirstlane((int)group_vec(gfirst));
        pcur.d = first_lane(desc_vec(pcur.g));
        pcur.pm = mask_vec(pcur.d);
        if (gfirst + 1u < gend) { g1 = (unsigned)__builtin_amdgcn_readfirstlane((int)group_vec(gfirst + 1u)); dv1 = desc_vec(g1); }
        if (gfirst + 2u < gend) gv2 = group_vec(gfirst + 2u);
    }
    // The loop of its own (SellDev::all_lists). Per slice: the descriptor (requested two slices ahead), the lists only when they
    // are not the previous slice's, N gathers of a compile-time N, N products; 85 vector + 134 scalar instructions per slice of
    // the 7-point Laplacian went through sell_row (rocprofv3: the scalar unit busy half of the time, waves waiting for memory a
    // fifth of theirs).
    if (PRE && !OFFD && lean && !done && gfirst < gend) {         // (done: nothing is stored and nothing published -- nothing to do)
        const double *__restrict__ x = a.x;
        const unsigned last = gend - 1u;                          // (requests past the workgroup's last group repeat it: no tests)
        uint32_t cy = 0xFFFFFFFFu, cz = 0xFFFFFFFFu;              // the lists in registers
        sell_i8 o8 = (sell_i8)(0);
        double lv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        const unsigned pw = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
        double *yp = a.y + (slice_of(pcur.g, pw) * kSliceRows + lane);   // where the previous slice's result goes (first slice: a zero
        double yv = 0.0;                                          // into its own row, overwritten by its result one slice later)
        for (unsigned gq = gfirst; gq < gend; ++gq) {
            SellPre pnext;
            pnext.g = g1; pnext.d = first_lane(dv1);              // (arrived during the previous slice)
            const unsigned g2 = (unsigned)__builtin_amdgcn_readfirstlane((int)gv2);
            pnext.pm = mask_vec(pnext.d);
            dv1 = desc_vec(g2);
            gv2 = group_vec(gq + 3u < last ? gq + 3u : last);
            const uint32_t row = slice_of(pcur.g, pw) * kSliceRows + lane;
            const uint32_t kind = pcur.d.x >> 16, len = pcur.d.x & 0xFFFFu;
            {                                                     // (no slice without rows: build_slice_desc -- a path that requests
                                                                  // nothing would make every wait of the loop a wait for everything)
                double upre = 0.0;
                if (NDOT >= 1) upre = a.u[row];
                if (pcur.d.y != cy || pcur.d.z != cz) {
                    cy = pcur.d.y; cz = pcur.d.z;
                    o8 = *(const BICG_KCONST sell_i8 *)(a.sell.uoff8 + cy);
                    const BICG_KCONST double *uv = (const BICG_KCONST double *)(a.sell.uval + cz);
#pragma unroll
                    for (int e = 0; e < 8; ++e) lv[e] = uv[e];
                }
                const uint32_t boff = row << 3;
                const double sum = kind == (uint32_t)kSliceMasked ? sell_list_switch<true>(len, x, boff, o8, lv, pcur.pm, yp, yv)
                                                                  : sell_list_switch<false>(len, x, boff, o8, lv, 0u, yp, yv);
                const double yi = 0.0 + sum;                      // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
                yp = a.y + row; yv = yi;
                if (NDOT >= 1) {
                    acc[0] += upre * yi;
                    if (NDOT == 2) acc[NDOT >= 2 ? 1 : 0] += yi * yi;
                    if (NDOT == 3) acc[NDOT >= 2 ? 1 : 0] += upre * upre;
                }
            }
            pcur = pnext; g1 = g2;
        }
        *yp = yv;
    }
    for (unsigned gq = gfirst; gq < gend && !(PRE && !OFFD && lean); gq += LL ? nblocks : 1u) {
        const unsigned gi = (a.reverse && !LL) ? gfirst + (gend - 1u - gq) : gq;      // a reversed product walks its groups backwards too
        uint32_t row;
        bool live;
        if (LAY == LAY_JAGW) sell_stage_window(a, a.glist ? a.glist[gi] : gi, dyn_lds);
        SellPre pnext = pcur;
        unsigned g2 = 0u;
        if (PRE && pre_on) {
            pnext.g = g1; pnext.d = first_lane(dv1);              // (arrived during the previous slice)
            g2 = (unsigned)__builtin_amdgcn_readfirstlane((int)gv2);
            pnext.pm = gq + 1u < gend ? mask_vec(pnext.d) : 0u;
            if (gq + 2u < gend) dv1 = desc_vec(g2);
            if (gq + 3u < gend) gv2 = group_vec(gq + 3u);
        }
        // the dot operand of this lane's row is requested BEFORE the row product (one load in front of the product's
        // batches; after it, it was a dependent round trip at the very end of every workgroup)
        double upre = 0.0;
        const uint32_t rguess = ((PRE && pre_on) ? pcur.g : (a.glist ? a.glist[gi] : gi)) * kGroupRows + threadIdx.x;
        if (NDOT >= 1 && rguess < a.nrows) upre = a.u[rguess];
        const double yi = sell_row<OFFD, NT, LAY, LL>(a, gi, done, row, live, ll_failed, dyn_lds, PRE && pre_on, pcur);
        if (PRE && pre_on) { pcur = pnext; g1 = g2; }
        if (live && !done) a.y[row] = yi;
        if (NDOT >= 1 && live) {
            const double ume = row == rguess ? upre : a.u[row];
            acc[0] += ume * yi;
            if (NDOT == 2) acc[NDOT >= 2 ? 1 : 0] += yi * yi;
            if (NDOT == 3) acc[NDOT >= 2 ? 1 : 0] += ume * ume;
        }
    }
    if (LL && ll_failed) { a.S->comm_error = 1; a.S->done = 1; }
    if (NDOT > 0 && !done) {
        if (MODE == RED_WAVE) wave_publish<(NDOT > 0 ? NDOT : 1)>(acc, a.red.partial, a.red.slot_base + slot);
        else reduce_publish<(NDOT > 0 ? NDOT : 1), MODE == RED_TICKET_HEAVY>(acc, a.S, a.red, a.red.slot_base + slot, sm, a.red.slot_base + bid);
    }
}

// ------------------------------------------------------------------------------------------
// SpMV + element-wise phase in ONE launch: the pipelined iteration as two kernels
//   EPI = 1:  v = A z ; then phase 2 on the workgroup's own rows (x, r, w, five dots; src/solver.c:366-380)
//   EPI = 2:  t = A w ; then phase 1 of the NEXT iteration (p, s, z, q, y, two dots; src/solver.c:352-364)
// The phase needs, per row, only values of that row -- among them the y_i this lane has just
// computed -- so it rides in the SpMV's epilogue: two launches per iteration instead of four, and the
// dot group the phase's scalars come from (produced by the previous launch) is summed by this
// launch's first workgroups while everybody streams the matrix; by the time a workgroup reaches its
// epilogue the totals are there. On a 200 k-row rank (1/8 of Transport) every launch boundary and
// every exposed reduction costs as much as the arithmetic, which is what this removes.
// The phase's expressions are those of FPipe1 / FPipe2, operation for operation; y lives in its own
// vector (a.epi.y) because w is this SpMV's input while the epilogue of EPI = 2 produces y.
// ------------------------------------------------------------------------------------------
template <int EPI, bool OFFD, bool NT, int LAY, bool LL>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) k_spmv_sell_epi(SpmvArgs a)
{
    constexpr int ND = EPI == 1 ? 5 : 2;
    const int done = a.S->done;
    __shared__ FinishLds fl;
    __shared__ Scal priv;
    unsigned bid = blockIdx.x, nblocks = gridDim.x;
    bool ll_failed = false;
    if (LL) {
        if (bid < a.ll.npush) { sell_halo_push(a, bid, done); return; }
        bid -= a.ll.npush; nblocks -= a.ll.npush;
    }
    // The open dot group. kShards DEDICATED workgroups in front of the row workgroups sum the shards (and,
    // peer-to-peer, the first of them hands the local sums to the other ranks) and leave; the first one also applies
    // the recurrence and writes the next scalar block. A workgroup with rows of its own would start them that much
    // later, and the launch ends with its slowest workgroup. Row workgroups apply privately at their epilogue.
    const unsigned nhelp = a.fin.seq && (a.fin.roles & FIN_SHARDS) ? (unsigned)kShards : 0u;
    if (bid < nhelp) {
        (void)finish_group(a.S, a.fin, a.fin.roles & (FIN_SHARDS | FIN_PUSH), bid, nhelp, fl, nullptr);
        if (bid == 0) {
            // ... applies the recurrence, writes the next scalar block, and hands the few scalars the phase needs to the
            // row workgroups as LL words (row kShards of the shard-total table): one small poll at their epilogue
            // instead of kShards x n totals, a reduction and the recurrence in every workgroup
            (void)finish_group(a.S, a.fin, FIN_APPLY, 0u, nhelp, fl, &priv);
        }
        return;
    }
    bid -= nhelp; nblocks -= nhelp;
    // staged by an earlier launch (or nothing to sum): row workgroup 0 publishes the scalar block
    const unsigned fin_bid = nhelp ? bid + 1u : bid;
    double acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = 0.0;
    const Vecs &e = a.epi;
    bool have = false;
    int sdone = done;
    double alpha = 0.0, beta = 0.0, omega = 0.0;
#define EPI_SCALARS()                                                                                     \
    do {                                                                                                  \
        const Scal *sc_ = a.S;                                                                            \
        bool got_ = false;                                                                                \
        if (done) {                        /* converged before this launch: nothing is published, nothing changes */ \
            got_ = true; sdone = 1;                                                                       \
        } else if (a.fin.seq && nhelp) {   /* the scalars as published by helper workgroup 0 */           \
            if (threadIdx.x == 0) fl.missing = 0u;                                                        \
            __syncthreads();                                                                              \
            if (threadIdx.x < 4) {                                                                        \
                double v_;                                                                                \
                const llword *w_ = a.fin.shard + ((size_t)kShards * kRedSlots + threadIdx.x) * 2;         \
                if (a.fin.p2p.seq ? ll_wait_agent(w_, a.fin.seq, a.fin.p2p.timeout_ticks, &v_)            \
                                  : ll_try_agent(w_, a.fin.seq, a.fin.spin_ticks, &v_))                   \
                    fl.sums[threadIdx.x] = v_;                                                            \
                else atomicOr(&fl.missing, 1u);                                                           \
            }                                                                                             \
            __syncthreads();                                                                              \
            got_ = fl.missing == 0u;                                                                      \
            if (got_) { alpha = fl.sums[0]; beta = fl.sums[1]; omega = fl.sums[2]; sdone = fl.sums[3] != 0.0 ? 1 : 0; } \
            __syncthreads();                                                                              \
        }                                                                                                 \
        if (!got_) {                       /* staged earlier, nothing open, or helper 0 is late: apply here */ \
            if (a.fin.seq) sc_ = finish_group(a.S, a.fin, FIN_APPLY, fin_bid, nblocks, fl, &priv);        \
            sdone = sc_->done; alpha = sc_->alpha; beta = sc_->beta; omega = sc_->omega;                  \
        }                                                                                                 \
        have = true;                                                                                      \
    } while (0)
    unsigned slot = bid;
    for (unsigned gi0 = bid; gi0 < a.nlist; gi0 += nblocks) {
        const unsigned gi = (a.reverse && !LL) ? a.nlist - 1u - gi0 : gi0;
        if (nblocks == a.nlist) slot = gi;
        uint32_t row;
        bool live;
        if (LAY == LAY_JAGW) sell_stage_window(a, a.glist ? a.glist[gi] : gi, dyn_lds);
        const double yi = sell_row<OFFD, NT, LAY, LL, 8>(a, gi, done, row, live, ll_failed, dyn_lds);
        if (live && !done) a.y[row] = yi;
        // The phase's inputs (values of this lane's own row) are requested right after the row product and BEFORE the
        // scalars are waited for: their round trip and the scalar poll's are one. (Requested before the row product
        // they sit in front of its loads -- loads return in issue order -- and delay every batch: BICG_EPI_EARLY.)
        const uint32_t rr_ = live ? row : 0u;
        double in0, in1, in2, in3, in4, in5, in6 = 0.0, in7 = 0.0;
        if (EPI == 1) { in0 = e.r[rr_]; in1 = e.y[rr_]; in2 = e.x[rr_]; in3 = e.p[rr_]; in4 = e.t[rr_]; in5 = e.rh[rr_]; in6 = e.s[rr_]; in7 = e.z[rr_]; }
        else { in0 = e.r[rr_]; in1 = e.w[rr_]; in2 = e.s[rr_]; in3 = e.z[rr_]; in4 = e.p[rr_]; in5 = e.v[rr_]; }
        if (!have) EPI_SCALARS();
        if (EPI == 1) {
            const double q = in0, y = in1, x0 = in2, p0 = in3, t0 = in4, h = in5, s0 = in6, z0 = in7;
            if (live && !sdone) {
                double xx = x0 + alpha * p0;
                xx = xx + omega * q;
                e.x[row] = xx;
                const double rr = q + (-omega) * y;
                e.r[row] = rr;
                const double tt = t0 + (-alpha) * yi;
                const double ww = y + (-omega) * tt;
                e.w[row] = ww;
                acc[0] += rr * rr; acc[1] += h * rr; acc[2] += h * ww; acc[3] += h * s0; acc[4] += h * z0;
            }
        } else {
            const double r0 = in0, w0 = in1, s0 = in2, z0 = in3, p0 = in4, v0 = in5;
            if (live && !sdone) {
                e.p[row] = recur3<double>(p0, s0, r0, omega, beta);
                const double s1 = recur3<double>(s0, z0, w0, omega, beta);
                const double z1 = recur3<double>(z0, v0, yi, omega, beta);
                e.s[row] = s1; e.z[row] = z1;
                const double q = r0 + (-alpha) * s1;
                const double y = w0 + (-alpha) * z1;
                e.r[row] = q; e.y[row] = y;
                acc[0] += q * y; acc[1] += y * y;
            }
        }
    }
    if (!have && fin_bid == 0) EPI_SCALARS();  // the publishing workgroup writes the scalar block even without rows
#undef EPI_SCALARS
    if (LL && ll_failed) { a.S->comm_error = 1; a.S->done = 1; }
    if (have && !sdone) wave_publish<ND>(acc, a.red.partial, a.red.slot_base + slot);
    else if (!have && !done) wave_publish<ND>(acc, a.red.partial, a.red.slot_base + slot);
}


static inline int sell_layout(const SellDev &d)
{
    if (d.win_slots) return LAY_JAGW;
    if (!d.jag && d.vbase) return d.col16 ? LAY_PAD16C : LAY_PAD32C;
    return (d.jag ? LAY_JAG32 : LAY_PAD32) + (d.col16 ? 1 : 0);
}

// workgroups launched for ngroups 256-row groups: every workgroup gets the same number (+-1)
#if PART_IS(0)
unsigned sell_grid(uint32_t ngroups, int per_wg)
{
    if (ngroups == 0) return 0;
    if (per_wg < 1) per_wg = 1;
    const unsigned per = (unsigned)per_wg;
    const unsigned grid0 = (ngroups + per - 1) / per;             // upper bound on workgroups
    const unsigned each = (ngroups + grid0 - 1) / grid0;
    return (ngroups + each - 1) / each;
}
#endif

// One sliced-ELL layout's instantiations (96 SpMV kernels + 12 with an epilogue): a translation unit each.
template <int LAY>
static bool sell_launch_layout(const SpmvArgs &a, int ndot, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1, bool fused_halo)
{
    if (a.nlist == 0 && !(fused_halo && a.ll.npush > 0)) return false;
    dim3 g(sell_grid(a.nlist, a.groups_per_wg) + (fused_halo ? a.ll.npush : 0u)), b(kBlock);
    const unsigned lds = LAY == LAY_JAGW ? a.sell.win_slots * (unsigned)sizeof(double) : 0u;
#define SELL_MODE(ND, OF, LLV, MD)                                                                 \
    do {                                                                                           \
        if (nt) launch_timed_lds(k_spmv_sell<ND, OF, true, LAY, LLV, MD>, g, b, lds, st, e0, e1, a);   \
        else launch_timed_lds(k_spmv_sell<ND, OF, false, LAY, LLV, MD>, g, b, lds, st, e0, e1, a);     \
    } while (0)
#define SELL_CASE(ND, OF, LLV)                                                                     \
    do {                                                                                           \
        const bool nt = a.nt != 0;                                                                 \
        const int mode = red_mode(a.red, a.fin, (ND) > 0);                                         \
        constexpr int HV = (ND) > 0 ? RED_TICKET_HEAVY : RED_TICKET;                               \
        if (mode == RED_WAVE) SELL_MODE(ND, OF, LLV, RED_WAVE);                                    \
        else if (mode == RED_TICKET_HEAVY) SELL_MODE(ND, OF, LLV, HV);                             \
        else SELL_MODE(ND, OF, LLV, RED_TICKET);                                                   \
    } while (0)
    if (fused_halo) {
        if (ndot == 0) SELL_CASE(0, true, true); else if (ndot == 1) SELL_CASE(1, true, true); else if (ndot == 2) SELL_CASE(2, true, true); else SELL_CASE(3, true, true);
    } else if (with_offd) {
        if (ndot == 0) SELL_CASE(0, true, false); else if (ndot == 1) SELL_CASE(1, true, false); else if (ndot == 2) SELL_CASE(2, true, false); else SELL_CASE(3, true, false);
    } else {
        if (ndot == 0) SELL_CASE(0, false, false); else if (ndot == 1) SELL_CASE(1, false, false); else if (ndot == 2) SELL_CASE(2, false, false); else SELL_CASE(3, false, false);
    }
#undef SELL_CASE
#undef SELL_MODE
    return true;
}

template <int LAY>
static bool sell_epi_launch_layout(const SpmvArgs &a, int epi, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1, bool fused_halo)
{
    if (a.nlist == 0 && !(fused_halo && a.ll.npush > 0)) return false;
    const unsigned nhelp = a.fin.seq && (a.fin.roles & FIN_SHARDS) ? (unsigned)kShards : 0u;      // dedicated shard summers
    dim3 g(sell_grid(a.nlist, a.groups_per_wg) + (fused_halo ? a.ll.npush : 0u) + nhelp), b(kBlock);
    const bool nt = a.nt != 0;
    const unsigned lds = LAY == LAY_JAGW ? a.sell.win_slots * (unsigned)sizeof(double) : 0u;
#define EPI_CASE(EP, OF, LLV)                                                                      \
    do {                                                                                           \
        if (nt) launch_timed_lds(k_spmv_sell_epi<EP, OF, true, LAY, LLV>, g, b, lds, st, e0, e1, a);   \
        else launch_timed_lds(k_spmv_sell_epi<EP, OF, false, LAY, LLV>, g, b, lds, st, e0, e1, a);     \
    } while (0)
    if (epi == 1) {
        if (fused_halo) EPI_CASE(1, true, true); else if (with_offd) EPI_CASE(1, true, false); else EPI_CASE(1, false, false);
    } else {
        if (fused_halo) EPI_CASE(2, true, true); else if (with_offd) EPI_CASE(2, true, false); else EPI_CASE(2, false, false);
    }
#undef EPI_CASE
    return true;
}

#define SELL_PART_ARGS const SpmvArgs &a, int n, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1, bool fused_halo
bool launch_spmv_sell_pad32(SELL_PART_ARGS);
bool launch_spmv_sell_pad16(SELL_PART_ARGS);
bool launch_spmv_sell_jag32(SELL_PART_ARGS);
bool launch_spmv_sell_jag16(SELL_PART_ARGS);
bool launch_spmv_sell_epi_pad32(SELL_PART_ARGS);
bool launch_spmv_sell_epi_pad16(SELL_PART_ARGS);
bool launch_spmv_sell_epi_jag32(SELL_PART_ARGS);
bool launch_spmv_sell_epi_jag16(SELL_PART_ARGS);
bool launch_spmv_sell_jagw(SELL_PART_ARGS);
bool launch_spmv_sell_pad32c(SELL_PART_ARGS);
bool launch_spmv_sell_pad16c(SELL_PART_ARGS);
bool launch_spmv_sell_epi_pad32c(SELL_PART_ARGS);
bool launch_spmv_sell_epi_pad16c(SELL_PART_ARGS);
bool launch_spmv_sell_epi_jagw(SELL_PART_ARGS);
#if PART_IS(1)
bool launch_spmv_sell_pad32(SELL_PART_ARGS) { return sell_launch_layout<LAY_PAD32>(a, n, with_offd, st, e0, e1, fused_halo); }
bool launch_spmv_sell_epi_pad32(SELL_PART_ARGS) { return sell_epi_launch_layout<LAY_PAD32>(a, n, with_offd, st, e0, e1, fused_halo); }
#endif
#if PART_IS(2)
bool launch_spmv_sell_pad16(SELL_PART_ARGS) { return sell_launch_layout<LAY_PAD16>(a, n, with_offd, st, e0, e1, fused_halo); }
bool launch_spmv_sell_epi_pad16(SELL_PART_ARGS) { return sell_epi_launch_layout<LAY_PAD16>(a, n, with_offd, st, e0, e1, fused_halo); }
#endif
#if PART_IS(3)
bool launch_spmv_sell_jag32(SELL_PART_ARGS) { return sell_launch_layout<LAY_JAG32>(a, n, with_offd, st, e0, e1, fused_halo); }
bool launch_spmv_sell_epi_jag32(SELL_PART_ARGS) { return sell_epi_launch_layout<LAY_JAG32>(a, n, with_offd, st, e0, e1, fused_halo); }
#endif
#if PART_IS(4)
bool launch_spmv_sell_jag16(SELL_PART_ARGS) { return sell_launch_layout<LAY_JAG16>(a, n, with_offd, st, e0, e1, fused_halo); }
bool launch_spmv_sell_epi_jag16(SELL_PART_ARGS) { return sell_epi_launch_layout<LAY_JAG16>(a, n, with_offd, st, e0, e1, fused_halo); }
#endif
#if PART_IS(5)
bool launch_spmv_sell_jagw(SELL_PART_ARGS) { return sell_launch_layout<LAY_JAGW>(a, n, with_offd, st, e0, e1, fused_halo); }
bool launch_spmv_sell_epi_jagw(SELL_PART_ARGS) { return sell_epi_launch_layout<LAY_JAGW>(a, n, with_offd, st, e0, e1, fused_halo); }
#endif
#if PART_IS(6)
bool launch_spmv_sell_pad32c(SELL_PART_ARGS) { return sell_launch_layout<LAY_PAD32C>(a, n, with_offd, st, e0, e1, fused_halo); }
bool launch_spmv_sell_epi_pad32c(SELL_PART_ARGS) { return sell_epi_launch_layout<LAY_PAD32C>(a, n, with_offd, st, e0, e1, fused_halo); }
#endif
#if PART_IS(7)
bool launch_spmv_sell_pad16c(SELL_PART_ARGS) { return sell_launch_layout<LAY_PAD16C>(a, n, with_offd, st, e0, e1, fused_halo); }
bool launch_spmv_sell_epi_pad16c(SELL_PART_ARGS) { return sell_epi_launch_layout<LAY_PAD16C>(a, n, with_offd, st, e0, e1, fused_halo); }
#endif
#undef SELL_PART_ARGS

// ---- code objects loaded at set-up, not at the first launch -------------------------------------------------------------
// The runtime loads a translation unit's code object when the first of its kernels is looked up: 10-80 ms for a unit with a few
// hundred sliced-ELL instantiations, paid in the MIDDLE of a solve whenever a kernel of a unit not used so far comes up (the
// first replacement step of pipe_bicgstab_rr, the first product with two dots, a leg of bench.py that follows a leg with
// another layout -- the "queue stall" of rounds 2-3). bicg_create looks up one kernel of every unit its context can launch
// from; BICG_PRELOAD=0 leaves the loading to the first launch.
template <int LAY> static void preload_layout()
{
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k_spmv_sell<0, false, false, LAY, false, RED_TICKET>));
    (void)hipGetLastError();
}
void preload_part1(); void preload_part2(); void preload_part3(); void preload_part4(); void preload_part5(); void preload_part6(); void preload_part7();
#if PART_IS(1)
void preload_part1() { preload_layout<LAY_PAD32>(); }
#endif
#if PART_IS(2)
void preload_part2() { preload_layout<LAY_PAD16>(); }
#endif
#if PART_IS(3)
void preload_part3() { preload_layout<LAY_JAG32>(); }
#endif
#if PART_IS(4)
void preload_part4() { preload_layout<LAY_JAG16>(); }
#endif
#if PART_IS(5)
void preload_part5() { preload_layout<LAY_JAGW>(); }
#endif
#if PART_IS(6)
void preload_part6() { preload_layout<LAY_PAD32C>(); }
#endif
#if PART_IS(7)
void preload_part7() { preload_layout<LAY_PAD16C>(); }
#endif
#if PART_IS(0)
void preload_kernels(const SellDev &d, bool sell)
{
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k_apply));      // this unit: element-wise kernels, CSR products, SpMM
    (void)hipGetLastError();
    if (!sell) return;
    switch (sell_layout(d)) {
    case LAY_PAD16: preload_part2(); break;
    case LAY_JAG32: preload_part3(); break;
    case LAY_JAG16: preload_part4(); break;
    case LAY_JAGW:  preload_part5(); break;
    case LAY_PAD32C: preload_part6(); break;
    case LAY_PAD16C: preload_part7(); break;
    default:        preload_part1(); break;
    }
}
#endif

#if PART_IS(0)
bool launch_spmv_sell(const SpmvArgs &a, int ndot, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1, bool fused_halo)
{
    if (a.nlist) {
        const int lay = sell_layout(a.sell);
        g_product_kernels |= lay == LAY_JAGW ? (jagw_fast_ok(a, with_offd, fused_halo) ? 0u : (unsigned)PK_SELL_WINLOOP)
                             : (lay == LAY_JAG32 || lay == LAY_JAG16) ? (jagd_fast_ok(a, with_offd, fused_halo) ? 0u : (unsigned)PK_SELL_JAG)
                             : (unsigned)PK_SELL_PAD;
    }
    switch (sell_layout(a.sell)) {
    case LAY_PAD16: return launch_spmv_sell_pad16(a, ndot, with_offd, st, e0, e1, fused_halo);
    case LAY_JAG32:
        if (jagd_fast_ok(a, with_offd, fused_halo)) return launch_spmv_jagd(a, ndot, st, e0, e1);
        return launch_spmv_sell_jag32(a, ndot, with_offd, st, e0, e1, fused_halo);
    case LAY_JAG16:
        if (jagd_fast_ok(a, with_offd, fused_halo)) return launch_spmv_jagd(a, ndot, st, e0, e1);
        return launch_spmv_sell_jag16(a, ndot, with_offd, st, e0, e1, fused_halo);
    case LAY_JAGW:
        if (jagw_fast_ok(a, with_offd, fused_halo)) return launch_spmv_jagw(a, ndot, st, e0, e1);
        return launch_spmv_sell_jagw(a, ndot, with_offd, st, e0, e1, fused_halo);
    case LAY_PAD32C: return launch_spmv_sell_pad32c(a, ndot, with_offd, st, e0, e1, fused_halo);
    case LAY_PAD16C: return launch_spmv_sell_pad16c(a, ndot, with_offd, st, e0, e1, fused_halo);
    default:        return launch_spmv_sell_pad32(a, ndot, with_offd, st, e0, e1, fused_halo);
    }
}

bool launch_spmv_sell_epi(const SpmvArgs &a, int epi, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1, bool fused_halo)
{
    if (a.nlist) g_product_kernels |= PK_SELL_EPI;
    switch (sell_layout(a.sell)) {
    case LAY_PAD16: return launch_spmv_sell_epi_pad16(a, epi, with_offd, st, e0, e1, fused_halo);
    case LAY_JAG32: return launch_spmv_sell_epi_jag32(a, epi, with_offd, st, e0, e1, fused_halo);
    case LAY_JAG16: return launch_spmv_sell_epi_jag16(a, epi, with_offd, st, e0, e1, fused_halo);
    case LAY_JAGW:  return launch_spmv_sell_epi_jagw(a, epi, with_offd, st, e0, e1, fused_halo);
    case LAY_PAD32C: return launch_spmv_sell_epi_pad32c(a, epi, with_offd, st, e0, e1, fused_halo);
    case LAY_PAD16C: return launch_spmv_sell_epi_pad16c(a, epi, with_offd, st, e0, e1, fused_halo);
    default:        return launch_spmv_sell_epi_pad32(a, epi, with_offd, st, e0, e1, fused_halo);
    }
}
#endif

// ------------------------------------------------------------------------------------------
// Sliced-ELL SpMM: Y_j = (A + sigma_j I) X_j for kSpmmCols vectors at once -- the verification loop of
// the reference's shifted driver (src/test_shifted.c:129-154: one SpMV per shift, A read nsig
// times) with A read ONCE. X is held row-major, kSpmmCols values per row = one 128-byte line.
// A wavefront works on 8 rows at a time, 8 lanes per row, each lane owning TWO columns: the load of
// the X values of one matrix entry is then one instruction over 8 fully used lines (a first version
// with lane = row touched 64 lines per instruction, 16 bytes of each: 0.56 ms on Transport, bound by
// the vector L1's tag rate). val / col are read from memory once per wavefront, lane = row, fully
// coalesced, into LDS; the 8 lanes of a row then take them from there with broadcast reads (a
// version in which they loaded the same word from memory issued 8 x the load instructions of the
// SpMV: 469 us). Every column of every row is accumulated in stored order like mult() (reference
// src/matrix.c:506-515), so each Y_j is bit-identical to the SpMV of that column. With b given,
// || b - Y_j ||^2 is fused (workgroup sums go to partial[wg][col], k_colsum adds them in a fixed
// order) and Y is never written.
// ------------------------------------------------------------------------------------------
#if PART_IS(0)
template <int LAY, bool OFFD>
__global__ void __launch_bounds__(kBlock) k_spmm_sell(SpmmArgs a)
{
    // LAY: the block's sliced-ELL layout (SellLayout). Padded slices: entry k of lane l at base + k * 64 + l. Jagged slices
    // (ragged rows): step k holds the entries of the rows longer than k only -- the staging pass finds a lane's entry with
    // a ballot like sell_row does. With x windows the stored 16-bit value is an LDS slot of the SpMV's window: the column
    // comes back through the group's runs. Rows dealt to the lanes by decreasing length (SellDev::perm): the lane -> row
    // map of the slice goes through LDS.
    constexpr bool WIN = LAY == LAY_JAGW, C16 = (LAY & 1) != 0 || WIN, JAG = LAY >= LAY_JAG32;
    constexpr int NB = kSpmmCols;
    constexpr int KC = 16;                                   // matrix entries per row staged in LDS at a time
    static_assert(NB == 16, "8 lanes per row x 2 columns per lane");
    struct Ent { double v; uint32_t off, pad; };            // value and byte offset of the row of X it multiplies
    __shared__ Ent se[kBlock / 64][KC][kSliceRows];          // this wavefront's slice, entry-major: read from memory ONCE,
                                                             // lane = row, fully coalesced
    __shared__ double sm[(kBlock / 64) * NB];
    __shared__ uint32_t rowmap[kBlock];                      // row of slice lane l (identity without SellDev::perm)
    __shared__ uint2 wruns[WIN ? 64 : 1];                    // the group's window runs (WIN)
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned sub = lane >> 3, cp = lane & 7u;          // row within an 8-row batch, column pair
    // XCD-contiguous mapping: workgroup b runs on XCD b % 8 (observed placement, used for speed only), so
    // giving XCD x the x-th eighth of the row groups makes one L2 fetch (almost) every line of X once
    // instead of all eight fetching all of it (16 vectors: 8 x 205 MB on Transport)
    unsigned g = blockIdx.x;
    if (a.xcd_map) {
        const unsigned per = (a.ngroups + 7u) / 8u;
        g = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    double acc0 = 0.0, acc1 = 0.0;
    unsigned nwr = 0;
    if (g < a.ngroups) {
        rowmap[tid] = g * kGroupRows + ((WIN && a.sell.perm) ? (uint32_t)a.sell.perm[(size_t)g * kGroupRows + tid] : tid);
        if (WIN) {
            const uint32_t r0 = a.sell.win_ptr[g];
            nwr = a.sell.win_ptr[g + 1] - r0;
            if (tid < nwr && tid < 64u) wruns[tid] = a.sell.win_runs[r0 + tid];
        }
    }
    __syncthreads();
    if (g < a.ngroups) {
        const uint32_t slice = g * (kGroupRows / kSliceRows) + wave;      // this wavefront's 64 rows
        uint32_t base = 0u, len = 0u, base16 = 0u;
        if (slice * kSliceRows < a.nrows) {
            base = a.sell.slice_base[slice]; len = a.sell.slice_len[slice];
            if (C16 && !JAG) base16 = a.sell.slice_base16[slice];
        }
        const double sg0 = a.sigma ? a.sigma[2 * cp] : 0.0, sg1 = a.sigma ? a.sigma[2 * cp + 1] : 0.0;
        const char *const xb = reinterpret_cast<const char *>(a.xt) + 16u * cp;     // this lane's two columns
        // stage role: lane = row of the slice
        const uint32_t srow = rowmap[wave * kSliceRows + lane];
        const uint32_t srb = srow < a.nrows ? srow : 0u;
        const uint32_t slen_me = srow < a.nrows ? a.dptr[srow + 1] - a.dptr[srow] : 0u;
        uint32_t jpos = base;                                // jagged: first entry of the current step (wave-uniform)
        // shortest row of the slice: entries below it need no per-row test (the usual case is all of them)
        uint32_t minlen = slen_me;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(minlen, off, 64); minlen = o < minlen ? o : minlen; }
        // compute role: 8 batches of 8 rows; this lane's row in batch bt is bt * 8 + sub
        double s0[kSliceRows / 8], s1[kSliceRows / 8];
        uint32_t mylen[kSliceRows / 8];
#pragma unroll
        for (int bt = 0; bt < kSliceRows / 8; ++bt) {
            const uint32_t row = rowmap[wave * kSliceRows + bt * 8 + sub];
            s0[bt] = 0.0; s1[bt] = 0.0;
            mylen[bt] = row < a.nrows ? a.dptr[row + 1] - a.dptr[row] : 0u;
        }
        for (uint32_t k0 = 0; k0 < len; k0 += KC) {
            const uint32_t kn = len - k0 < (uint32_t)KC ? len - k0 : (uint32_t)KC;
            // ---- stage: coalesced loads, 512 bytes of val per instruction
            if (JAG) {
                for (uint32_t e = 0; e < kn; ++e) {
                    const bool mine = k0 + e < slen_me;
                    const unsigned long long m = __ballot(mine);
                    const uint32_t j = jpos + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    jpos += (uint32_t)__builtin_popcountll(m);
                    uint32_t colv = srb;
                    double vv = 0.0;
                    if (mine) {
                        vv = a.sell.val[j];
                        if (WIN) {
                            const uint32_t slot = reinterpret_cast<const unsigned short *>(a.sell.col16)[j];
                            unsigned r = 0;
                            while (r + 1 < nwr && (wruns[r + 1].y >> 16) <= slot) ++r;
                            colv = wruns[r].x + (slot - (wruns[r].y >> 16));
                        } else if (C16) {
                            colv = srb + (uint32_t)(int)a.sell.col16[j];
                        } else {
                            colv = a.sell.col[j];
                        }
                    }
                    se[wave][e][lane].off = colv * (NB * 8u);
                    se[wave][e][lane].v = vv;
                }
            } else if (C16) {
                for (uint32_t q = 0; 4 * q < kn; ++q) {
                    const i16x4 dq = *(reinterpret_cast<const i16x4 *>(a.sell.col16) + ((size_t)base16 / 4 + (size_t)(k0 / 4 + q) * kSliceRows + lane));
                    se[wave][4 * q + 0][lane].off = (srb + (int)dq.x) * (NB * 8u); se[wave][4 * q + 1][lane].off = (srb + (int)dq.y) * (NB * 8u);
                    se[wave][4 * q + 2][lane].off = (srb + (int)dq.z) * (NB * 8u); se[wave][4 * q + 3][lane].off = (srb + (int)dq.w) * (NB * 8u);
                }
            }
            for (uint32_t e = 0; !JAG && e < kn; ++e) {
                const uint32_t j = base + (k0 + e) * kSliceRows + lane;
                if (!C16) se[wave][e][lane].off = a.sell.col[j] * (NB * 8u);
                se[wave][e][lane].v = a.sell.val[j];
            }
            __builtin_amdgcn_wave_barrier();      // same wavefront writes and reads: program order of its LDS operations is enough
            // ---- multiply: per batch and entry one broadcast LDS read of {val, offset} and ONE load of 8 full lines
            // of X; the 8 batches are independent, their loads are in flight together. Entries below the slice's
            // shortest row take the path without per-row tests (the kernel is bound by instruction issue).
            const uint32_t nfast = k0 >= minlen ? 0u : (minlen - k0 < kn ? minlen - k0 : kn);
            for (uint32_t e = 0; e < nfast; ++e) {
                f64x2 x[kSliceRows / 8];
                double v[kSliceRows / 8];
#pragma unroll
                for (int bt = 0; bt < kSliceRows / 8; ++bt) {
                    const Ent en = se[wave][e][bt * 8 + sub];
                    v[bt] = en.v;
                    x[bt] = *reinterpret_cast<const f64x2 *>(xb + en.off);
                }
#pragma unroll
                for (int bt = 0; bt < kSliceRows / 8; ++bt) { s0[bt] += v[bt] * x[bt].x; s1[bt] += v[bt] * x[bt].y; }     // stored order
            }
            for (uint32_t e = nfast; e < kn; ++e) {
#pragma unroll
                for (int bt = 0; bt < kSliceRows / 8; ++bt) {
                    const Ent en = se[wave][e][bt * 8 + sub];
                    const f64x2 x = *reinterpret_cast<const f64x2 *>(xb + en.off);
                    if (k0 + e < mylen[bt]) { s0[bt] += en.v * x.x; s1[bt] += en.v * x.y; }                                 // padding never added
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int bt = 0; bt < kSliceRows / 8; ++bt) {
            const uint32_t row = rowmap[wave * kSliceRows + bt * 8 + sub];
            const bool live = row < a.nrows;
            double y0 = 0.0 + s0[bt], y1 = 0.0 + s1[bt];                  // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
            if (OFFD && live) {
                double o0 = 0.0, o1 = 0.0;
                for (uint32_t k = a.offd.ptr[row]; k < a.offd.ptr[row + 1]; ++k) {
                    const double v = a.offd.val[k];
                    const f64x2 x = *reinterpret_cast<const f64x2 *>(a.xt + (size_t)a.offd.col[k] * NB + 2 * cp);
                    o0 += v * x.x; o1 += v * x.y;
                }
                y0 += o0; y1 += o1;                                       // second mult() call, src/matrix.c:440
            }
            if (live) {
                if (a.sigma) {
                    const f64x2 x = *reinterpret_cast<const f64x2 *>(a.xt + (size_t)row * NB + 2 * cp);
                    y0 += sg0 * x.x; y1 += sg1 * x.y;                     // += sigma_j x_j (src/test_shifted.c:133)
                }
                if (a.yt) { f64x2 t; t.x = y0; t.y = y1; *reinterpret_cast<f64x2 *>(a.yt + (size_t)row * NB + 2 * cp) = t; }
                if (a.b) {
                    const double bi = a.b[row];
                    const double d0 = (bi + (-1.0) * y0) - 0.0, d1 = (bi + (-1.0) * y1) - 0.0;
                    acc0 += d0 * d0; acc1 += d1 * d1;
                }
            }
        }
    }
    if (a.b) {
        // lanes with the same column pair sit 8 apart: fold the 8 row positions, then the 4 wavefronts
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1) { acc0 += __shfl_down(acc0, off, 64); acc1 += __shfl_down(acc1, off, 64); }
        if (lane < 8) { sm[wave * NB + 2 * lane] = acc0; sm[wave * NB + 2 * lane + 1] = acc1; }
        __syncthreads();
        if (tid < NB) {
            double t = sm[tid];
            for (int w = 1; w < kBlock / 64; ++w) t += sm[w * NB + tid];
            a.partial[(size_t)blockIdx.x * NB + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Windowed SpMM (round 4): the same product with X read ONCE and no layout change. The row-major kernel above pulls a
// 128-byte line of X through the vector L1 for every matrix entry (2.4 GB per launch on Transport, 1.6 x the algorithmic
// bytes from memory, + 97 us of transposes). Here a workgroup owns a 256-row group like the SpMV does (lane = row), stages the
// x values the group touches for NV vectors at a time in LDS -- straight from the shift-major vectors, every run of
// consecutive columns one coalesced copy per vector, exactly what sell_stage_window does for one vector -- and then walks
// its rows once per pass: every matrix entry is loaded once per NV vectors and multiplies NV LDS reads (consecutive lanes
// read consecutive slots: conflict-free). Y goes back shift-major. Where the columns come from:
//   MODE 0  padded slices with 16-bit offsets (banded / stencil-like matrices): the offsets fall into <= 4 clusters (struct
//           FusedWindow), cluster k of every group is the run [g0 + lo_k, g0 + 255 + hi_k], slot = thread + offset + bias_k;
//   MODE 1  jagged slices with x windows (ragged rows): the stored 16-bit value IS the slot, the runs are the SpMV's.
// Per row and vector the sum runs in stored order like mult() (reference src/matrix.c:506-515): every column is bit-identical
// to bicg_spmv of that vector. NV = 8 on Transport (79 KB of LDS, two workgroups per CU); the first 16 entries of every row
// stay in registers across the passes, so the matrix is read once per launch whatever NV.
// ------------------------------------------------------------------------------------------
template <int MODE, bool OFFD, int NV>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 4))) k_spmm_win(SpmmArgs a)      // (the LDS window allows two workgroups per CU)
{
    constexpr bool WIN = MODE == 1;
    constexpr int U = 8;
    __shared__ double sm[(kBlock / 64) * NV];
    double *const win = dyn_lds;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    unsigned g = blockIdx.x;
    if (a.xcd_map) {
        const unsigned per = (a.ngroups + 7u) / 8u;
        g = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    if (a.b && tid < (unsigned)kSpmmCols && (g >= a.ngroups || (int)tid >= a.nvec)) a.partial[(size_t)blockIdx.x * kSpmmCols + tid] = 0.0;
    if (g >= a.ngroups) return;                               // (grid padded to a multiple of 8: workgroup-uniform)
    const unsigned W = a.wslots;
    const uint32_t g0 = g * kGroupRows;
    const uint32_t row = g0 + ((WIN && a.sell.perm) ? (uint32_t)a.sell.perm[(size_t)g0 + tid] : tid);
    const uint32_t slice = g * (kGroupRows / kSliceRows) + wave;
    const bool live = row < a.nrows;
    uint32_t base = 0u, len = 0u, base16 = 0u;
    if (slice * kSliceRows < a.nrows) {
        base = a.sell.slice_base[slice]; len = a.sell.slice_len[slice];
        if (!WIN) base16 = a.sell.slice_base16[slice];
    }
    const uint32_t mylen = live ? a.dptr[row + 1] - a.dptr[row] : 0u;
    uint32_t oa = 0u, ob = 0u;
    if (OFFD && live) { oa = a.offd.ptr[row]; ob = a.offd.ptr[row + 1]; }
    const double bi = (a.b && live) ? a.b[row] : 0.0;
    const unsigned short *const slots16 = reinterpret_cast<const unsigned short *>(a.sell.col16);
    __shared__ uint2 wruns[WIN ? 64 : 1];                    // the group's window runs (spmm_possible: at most 64)
    unsigned nwr = 0;
    // (round 6) LIST: the group's window is a list of its distinct columns (SellDev::win_list, the layout k_spmv_jagl stages from):
    // slot s = position in the list, column = the group's first row + a 16-bit distance -- no runs, any number of them
    const bool LIST = WIN && a.sell.win_list != nullptr;
    const uint32_t lbase = LIST ? a.sell.win_lptr[g] : 0u, ltotal = LIST ? a.sell.win_ltotal[g] : 0u;
    if (WIN && !LIST) {
        const uint32_t r0 = a.sell.win_ptr[g];
        nwr = a.sell.win_ptr[g + 1] - r0;
        if (tid < nwr && tid < 64u) wruns[tid] = a.sell.win_runs[r0 + tid];
    }

    // ---- the head of the row -- its first K entries, all of it for most matrices -- is loaded ONCE and kept in registers for
    // every pass (value, LDS slot, "counts" bit): the matrix is then read once per launch, not once per NV vectors
    constexpr int K = 16;
    double hv[K];
    unsigned hs[K];
    unsigned hon = 0u;
    uint32_t pos = base;                                      // jagged: first entry of step k (wave-uniform), as in sell_row
    auto slot_of = [&](int d) -> unsigned {                   // padded layout: slot = thread + distance + bias of the distance's cluster
        int bias = a.cl.bias[0];
        if (a.cl.ncl > 1 && d >= a.cl.lo[1]) bias = a.cl.bias[1];
        if (a.cl.ncl > 2 && d >= a.cl.lo[2]) bias = a.cl.bias[2];
        if (a.cl.ncl > 3 && d >= a.cl.lo[3]) bias = a.cl.bias[3];
        return (unsigned)((int)tid + d + bias);               // padding: distance 0, the row's own column
    };
    const i16x4 *const q16 = reinterpret_cast<const i16x4 *>(a.sell.col16) + ((size_t)base16 / 4 + lane);
    if (a.dbg & 4) {
#pragma unroll
        for (int e = 0; e < K; ++e) { hs[e] = tid; hv[e] = 1.0; hon |= 1u << e; }
    } else if (WIN) {
#pragma unroll
        for (int e = 0; e < K; ++e) {
            hs[e] = 0u; hv[e] = 0.0;
            if ((uint32_t)e < len) {                          // wave-uniform
                const bool mine = (uint32_t)e < mylen;
                const unsigned long long m = __ballot(mine);
                const uint32_t j = pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                pos += (uint32_t)__builtin_popcountll(m);
                if (mine) { hs[e] = slots16[j]; hv[e] = a.sell.val[j]; hon |= 1u << e; }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < K / 4; ++q) {
            i16x4 dq = (i16x4)(0);
            if ((uint32_t)(4 * q) < len) dq = q16[(size_t)q * kSliceRows];        // wave-uniform test; the quad is padded
            hs[4 * q + 0] = slot_of(dq.x); hs[4 * q + 1] = slot_of(dq.y); hs[4 * q + 2] = slot_of(dq.z); hs[4 * q + 3] = slot_of(dq.w);
        }
#pragma unroll
        for (int e = 0; e < K; ++e) {
            hv[e] = (uint32_t)e < len ? a.sell.val[base + (uint32_t)e * kSliceRows + lane] : 0.0;
            if ((uint32_t)e < mylen) hon |= 1u << e;
        }
    }
    const uint32_t pos_tail = pos;

    for (int v0 = 0; v0 < a.nvec; v0 += NV) {
        const int nv = a.nvec - v0 < NV ? a.nvec - v0 : NV;
        __syncthreads();                                      // the previous pass has finished reading the window (and sm)
        // ---- stage the group's window of vectors v0 .. v0 + nv - 1: JB slots per thread and round, all their NV values
        // requested before the first is stored (one dependent round trip per run and vector: 743 us per launch on Transport;
        // one slot per round: 430 us)
        constexpr int JB = 40 / NV;                           // Transport's 1 240 slots in ONE round: two waves per SIMD hide no second trip
        for (unsigned s0 = 0; s0 < W; s0 += JB * kBlock) {
            int c[JB];                                        // column of the slot; -1: unused slot / clipped by the matrix boundary
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const unsigned sl = s0 + (unsigned)j * kBlock + tid;
                c[j] = -1;
                if (sl < W) {
                    if (LIST) {
                        if (sl < ltotal) {
                            const uint32_t word = a.sell.win_list[lbase + (sl >> 9) * (unsigned)kGroupRows + (sl & 255u)];
                            c[j] = (int)g0 + (int)(short)((sl & 256u) ? word >> 16 : word & 0xFFFFu);
                        }
                    } else if (WIN) {
                        unsigned r = 0;
                        while (r + 1 < nwr && (wruns[r + 1].y >> 16) <= sl) ++r;      // runs are few and ordered by slot
                        const unsigned off = sl - (wruns[r].y >> 16);
                        if (nwr && off < (wruns[r].y & 0xFFFFu)) c[j] = (int)(wruns[r].x + off);
                    } else {
                        int k = 0;
                        if (a.cl.ncl > 1 && (int)sl >= a.cl.bias[1] + a.cl.lo[1]) k = 1;
                        if (a.cl.ncl > 2 && (int)sl >= a.cl.bias[2] + a.cl.lo[2]) k = 2;
                        if (a.cl.ncl > 3 && (int)sl >= a.cl.bias[3] + a.cl.lo[3]) k = 3;
                        const int bk = k == 0 ? a.cl.bias[0] : k == 1 ? a.cl.bias[1] : k == 2 ? a.cl.bias[2] : a.cl.bias[3];
                        const int cc = (int)g0 + (int)sl - bk;                      // slot = (column - g0) + bias_k
                        if (cc >= 0 && cc < (int)a.nrows) c[j] = cc;
                    }
                }
            }
            double t[JB][NV];
#pragma unroll
            for (int j = 0; j < JB; ++j) {
#pragma unroll
                for (int v = 0; v < NV; ++v) t[j][v] = (c[j] >= 0 && v < nv && !(a.dbg & 1)) ? a.xs[(size_t)(v0 + v) * a.vstride + (unsigned)c[j]] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const unsigned sl = s0 + (unsigned)j * kBlock + tid;
                if (sl < W) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) win[(unsigned)v * W + sl] = t[j][v];
                }
            }
        }
        __syncthreads();
        // ---- the rows, NV sums per lane: the head from registers, whatever follows streamed
        double acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.0;
        // Straight-line code per half of the head: with a (wave-uniform) branch around every entry the compiler waited for each
        // LDS read before issuing the next -- 120 exposed LDS latencies per pass and lane, 412 us per launch on Transport. Entries
        // past the slice's length read slot 0 and are never added.
        // (the slots are made opaque once per pass: otherwise the 16 x NV LDS addresses slot + v W are hoisted out of the pass
        // loop as invariants and held in 128 registers)
#pragma unroll
        for (int e = 0; e < K; ++e) asm volatile("" : "+v"(hs[e]));
        auto half = [&](int e0) {
#pragma unroll
            for (int e = e0; e < e0 + K / 2; ++e) {
                double xr[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) xr[v] = win[(unsigned)v * W + hs[e]];
                const bool on = (hon >> e) & 1u;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double t = acc[v] + hv[e] * xr[v];  // stored order; padding never added
                    acc[v] = on ? t : acc[v];
                }
                // (two entries = 16 reads in flight are enough; left alone the scheduler hoists all 64 of a half: 256 registers)
                if ((e & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (!(a.dbg & 2)) {
        half(0);
        if (len > (uint32_t)(K / 2)) half(K / 2);
        }
        pos = pos_tail;
        for (uint32_t k0 = K; k0 < len && !(a.dbg & 2); k0 += U) {
            double val[U];
            unsigned sl[U];
            bool on[U];
            if (WIN) {
#pragma unroll
                for (int e = 0; e < U; ++e) {
                    on[e] = k0 + e < mylen;
                    const unsigned long long m = __ballot(on[e]);
                    const uint32_t j = pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    pos += (uint32_t)__builtin_popcountll(m);
                    sl[e] = 0u; val[e] = 0.0;
                    if (on[e]) { sl[e] = slots16[j]; val[e] = a.sell.val[j]; }
                }
            } else {
#pragma unroll
                for (int q = 0; q < U / 4; ++q) {
                    i16x4 dq = (i16x4)(0);
                    if (k0 + 4 * q < len) dq = q16[(size_t)(k0 / 4 + q) * kSliceRows];
                    sl[4 * q + 0] = slot_of(dq.x); sl[4 * q + 1] = slot_of(dq.y); sl[4 * q + 2] = slot_of(dq.z); sl[4 * q + 3] = slot_of(dq.w);
                }
#pragma unroll
                for (int e = 0; e < U; ++e) {
                    val[e] = k0 + e < len ? a.sell.val[base + (k0 + e) * kSliceRows + lane] : 0.0;
                    on[e] = k0 + e < mylen;
                }
            }
#pragma unroll
            for (int e = 0; e < U; ++e) {
                double xr[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) xr[v] = win[(unsigned)v * W + sl[e]];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double t = acc[v] + val[e] * xr[v]; // stored order
                    acc[v] = on[e] ? t : acc[v];
                }
            }
        }
        double r2[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            r2[v] = 0.0;
            if (v < nv && live) {
                const double *xv = a.xs + (size_t)(v0 + v) * a.vstride;
                double y = 0.0 + acc[v];                      // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
                if (OFFD) {
                    double so = 0.0;
                    for (uint32_t k = oa; k < ob; ++k) so += a.offd.val[k] * xv[a.offd.col[k]];
                    y += so;                                  // second mult() call, src/matrix.c:440
                }
                // += sigma_j x_j (src/test_shifted.c:133); with clusters the row's own column is in the window (distance 0 is always
                // part of a cluster): no trip to memory at the end of the pass
                if (a.sigma) y += a.sigma[v0 + v] * (WIN ? xv[row] : win[(unsigned)v * W + slot_of(0)]);
                if (a.ys) a.ys[(size_t)(v0 + v) * a.vstride + row] = y;
                if (a.b) { const double dd = (bi + (-1.0) * y) - 0.0; r2[v] = dd * dd; }
            }
        }
        if (a.b) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double t = wave_sum(r2[v]);
                if (lane == 0) sm[wave * NV + v] = t;
            }
            __syncthreads();
            if ((int)tid < nv) {
                double t = sm[tid];
                for (int w = 1; w < kBlock / 64; ++w) t += sm[w * NV + tid];
                a.partial[(size_t)blockIdx.x * kSpmmCols + v0 + tid] = t;
            }
        }
    }
}


// out[col] = sum over workgroups of partial[wg][col], fixed order; one workgroup per column
__global__ void __launch_bounds__(kBlock) k_colsum(const double *partial, unsigned nwg, double *out)
{
    constexpr int NB = kSpmmCols;
    __shared__ double sm[5];
    const unsigned col = blockIdx.x;
    double t[1] = {0.0};
    for (unsigned w = threadIdx.x; w < nwg; w += kBlock) t[0] += partial[(size_t)w * NB + col];
    block_sum<1>(t, sm);
    if (threadIdx.x == 0) out[col] = t[0];
}

// shift-major vectors x[j * stride + i] <-> row-major xt[i * kSpmmCols + j] (columns >= nvec are zero), a tile of
// 256 rows through LDS so that both the reads and the writes are coalesced
__global__ void __launch_bounds__(kBlock) k_rows_from_vectors(const double *x, size_t stride, int nvec, uint32_t n, double *xt)
{
    constexpr int NB = kSpmmCols;
    __shared__ double tile[kBlock][NB + 1];
    const uint32_t r0 = blockIdx.x * kBlock, i = r0 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < NB; ++j) tile[threadIdx.x][j] = (j < nvec && i < n) ? x[(size_t)j * stride + i] : 0.0;
    __syncthreads();
    // piece q of the tile = 16 bytes: row q / 8, columns 2 (q % 8), +1; consecutive threads write consecutive pieces
    for (unsigned q = threadIdx.x; q < kBlock * (NB / 2); q += kBlock) {
        const unsigned row = q / (NB / 2), c2 = (q % (NB / 2)) * 2;
        if (r0 + row < n) {
            f64x2 t; t.x = tile[row][c2]; t.y = tile[row][c2 + 1];
            *reinterpret_cast<f64x2 *>(xt + (size_t)(r0 + row) * NB + c2) = t;
        }
    }
}
__global__ void __launch_bounds__(kBlock) k_vectors_from_rows(const double *yt, size_t stride, int nvec, uint32_t n, double *y)
{
    constexpr int NB = kSpmmCols;
    __shared__ double tile[kBlock][NB + 1];
    const uint32_t r0 = blockIdx.x * kBlock, i = r0 + threadIdx.x;
    for (unsigned q = threadIdx.x; q < kBlock * (NB / 2); q += kBlock) {
        const unsigned row = q / (NB / 2), c2 = (q % (NB / 2)) * 2;
        if (r0 + row < n) {
            const f64x2 t = *reinterpret_cast<const f64x2 *>(yt + (size_t)(r0 + row) * NB + c2);
            tile[row][c2] = t.x; tile[row][c2 + 1] = t.y;
        }
    }
    __syncthreads();
    if (i < n) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < nvec) y[(size_t)j * stride + i] = tile[threadIdx.x][j];
    }
}

void launch_spmm_sell(const SpmmArgs &a, bool with_offd, hipStream_t st)
{
    if (a.ngroups == 0) return;
    const unsigned grid = a.xcd_map ? ((a.ngroups + 7u) / 8u) * 8u : a.ngroups;
#define SPMM_GO(LAYV)                                                                                      \
    do {                                                                                                   \
        if (with_offd) BICG_LAUNCH((k_spmm_sell<LAYV, true>), dim3(grid), dim3(kBlock), 0, st, a);         \
        else BICG_LAUNCH((k_spmm_sell<LAYV, false>), dim3(grid), dim3(kBlock), 0, st, a);                  \
    } while (0)
    switch (sell_layout(a.sell)) {
    case LAY_PAD16: SPMM_GO(LAY_PAD16); break;
    case LAY_JAG32: SPMM_GO(LAY_JAG32); break;
    case LAY_JAG16: SPMM_GO(LAY_JAG16); break;
    case LAY_JAGW:  SPMM_GO(LAY_JAGW); break;
    default:        SPMM_GO(LAY_PAD32); break;
    }
#undef SPMM_GO
}
// vectors per window: as many as leave room for two workgroups per CU (80 KB each), else whatever fits one

int spmm_win_vectors(unsigned wslots)
{
    if (wslots == 0) return 0;
    static const int forced = knob_x("BICG_SPMM_NV") ? atoi(knob_x("BICG_SPMM_NV")) : 0;      // measurement knob: 4 or 8 vectors per window
    if ((forced == 4 || forced == 8) && (size_t)forced * wslots * 8u <= 156u * 1024u) return forced;
    // (the head of every row stays in registers across the passes, so more vectors per window save barriers, not matrix traffic:
    // 16 per window was dropped -- its 256 LDS reads per thread in flight cost the occupancy)
    for (int nv : {8, 4}) if ((size_t)nv * wslots * 8u <= 80u * 1024u) return nv;
    for (int nv : {8, 4}) if ((size_t)nv * wslots * 8u <= 156u * 1024u) return nv;
    return 0;
}
hipError_t launch_spmm_win(const SpmmArgs &a, bool with_offd, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    if (a.ngroups == 0) return hipSuccess;
    const int nv = spmm_win_vectors(a.wslots);
    if (!nv) return hipErrorInvalidValue;
    const unsigned grid = a.xcd_map ? ((a.ngroups + 7u) / 8u) * 8u : a.ngroups;
    const unsigned lds = (unsigned)nv * a.wslots * 8u;
    const bool runs = a.cl.ncl == 0;
    auto go = [&](auto kernel) {
        // (the runtime answers "invalid argument" and launches with > 64 KiB of dynamic LDS all the same: bicg_persist.hip)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        (void)hipGetLastError();
        if (e0 && e1) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, st, e0, e1, 0, a);
        else hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, st, a);
        return hipGetLastError();
    };
#define WIN_GO(NVV)                                                                                                   \
    (runs ? (with_offd ? go(k_spmm_win<1, true, NVV>) : go(k_spmm_win<1, false, NVV>))                                 \
          : (with_offd ? go(k_spmm_win<0, true, NVV>) : go(k_spmm_win<0, false, NVV>)))
    const hipError_t err = nv == 8 ? WIN_GO(8) : WIN_GO(4);
#undef WIN_GO
    return err;
}
unsigned spmm_grid(uint32_t ngroups, bool xcd_map) { return xcd_map ? ((ngroups + 7u) / 8u) * 8u : ngroups; }
void launch_colsum(const double *partial, unsigned nwg, double *out, hipStream_t st)
{
    BICG_LAUNCH(k_colsum, dim3(kSpmmCols), dim3(kBlock), 0, st, partial, nwg, out);
}
void launch_rows_from_vectors(const double *x, size_t stride, int nvec, uint32_t n, double *xt, hipStream_t st)
{
    if (n) BICG_LAUNCH(k_rows_from_vectors, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, x, stride, nvec, n, xt);
}
void launch_vectors_from_rows(const double *yt, size_t stride, int nvec, uint32_t n, double *y, hipStream_t st)
{
    if (n) BICG_LAUNCH(k_vectors_from_rows, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, yt, stride, nvec, n, y);
}
#endif

#if PART_IS(0)
void launch_apply(Scal *S, int phase, hipStream_t st)
{
    BICG_LAUNCH(k_apply, dim3(1), dim3(phase >= PH_SH_INIT ? kBlock : 1), 0, st, S, phase);
}

// gather the entries of x other ranks need into the contiguous send buffer
__global__ void __launch_bounds__(kBlock) k_halo_pack(const double *x, const uint32_t *idx, uint32_t n, double *out, const Scal *S)
{
    if (S->done) return;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = x[idx[i]];
}

void launch_halo_pack(const double *x, const uint32_t *send_idx, uint32_t nsend, double *sendbuf, Scal *S, hipStream_t st)
{
    if (nsend == 0) return;
    unsigned g = (nsend + kBlock - 1) / kBlock;
    if (g > 1024) g = 1024;
    BICG_LAUNCH(k_halo_pack, dim3(g), dim3(kBlock), 0, st, x, send_idx, nsend, sendbuf, S);
}

void launch_apply_p2p(Scal *S, int phase, int n, const P2pRed &pr, unsigned long long timeout_ticks, hipStream_t st)
{
    BICG_LAUNCH(k_apply_p2p, dim3(1), dim3(kBlock), 0, st, S, phase, n, pr, timeout_ticks);
}

__global__ void __launch_bounds__(kBlock) k_halo_push(const double *x, const uint32_t *idx, uint32_t n,
                                                      const unsigned long long *dst0, const unsigned long long *dst_stride,
                                                      unsigned seq, const Scal *S)
{
    if (S->done) return;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    llword *dst = reinterpret_cast<llword *>(dst0[i] + (unsigned long long)(seq % kHaloRing) * dst_stride[i]);
    ll_store(dst, x[idx[i]], seq);
}

void launch_halo_push(const double *x, const uint32_t *send_idx, uint32_t nsend, const unsigned long long *dst0,
                      const unsigned long long *dst_stride, unsigned seq, Scal *S, hipStream_t st)
{
    if (nsend == 0) return;
    BICG_LAUNCH(k_halo_push, dim3((nsend + kBlock - 1) / kBlock), dim3(kBlock), 0, st, x, send_idx, nsend, dst0,
                       dst_stride, seq, (const Scal *)S);
}

__global__ void __launch_bounds__(kBlock) k_halo_unpack(const llword *ring, uint32_t halo, unsigned seq, double *tail, Scal *S,
                                                        unsigned long long timeout_ticks)
{
    if (S->done) return;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= halo) return;
    double v;
    if (ll_wait(ring + ((size_t)(seq % kHaloRing) * halo + i) * 2, seq, timeout_ticks, &v)) {
        tail[i] = v;
    } else {
        S->comm_error = 1;
        S->done = 1;
    }
}

void launch_halo_unpack(const llword *ring, uint32_t halo, unsigned seq, double *tail, Scal *S,
                        unsigned long long timeout_ticks, hipStream_t st)
{
    if (halo == 0) return;
    BICG_LAUNCH(k_halo_unpack, dim3((halo + kBlock - 1) / kBlock), dim3(kBlock), 0, st, ring, halo, seq, tail, S,
                       timeout_ticks);
}

// Flow control for halo exchanges that are not separated by an all-reduce: every rank posts a token
// to every rank (slot kRedSlots-1 of the mailbox, its own sequence numbers) and waits for all P.
__global__ void __launch_bounds__(64) k_p2p_barrier(P2pRed pr, unsigned long long timeout_ticks, Scal *S)
{
    if (S->done) return;
    constexpr int d = kRedSlots - 1;
    for (int p = threadIdx.x; p < pr.nranks; p += 64)
        ll_store(pr.mail[p] + mail_index(pr.seq, pr.nranks, pr.rank, d), 0.0, pr.seq);
    for (int p = threadIdx.x; p < pr.nranks; p += 64) {
        double v;
        if (!ll_wait(pr.mail[pr.rank] + mail_index(pr.seq, pr.nranks, p, d), pr.seq, timeout_ticks, &v)) {
            S->comm_error = 1;
            S->done = 1;
        }
    }
}

void launch_p2p_barrier(const P2pRed &pr, unsigned long long timeout_ticks, Scal *S, hipStream_t st)
{
    BICG_LAUNCH(k_p2p_barrier, dim3(1), dim3(64), 0, st, pr, timeout_ticks, S);
}

// Second part of the transport self-test: the HALO pattern -- every rank stores `entries` values per
// round into the landing ring of every other rank and reads what the others stored into its own,
// for more rounds than the ring has slots (a reused slot must never be read with its old contents),
// with the ranks deliberately out of step and the solver's flow control (a token barrier every
// kHaloRing - 2 exchanges). Ring layout: [kHaloRing][source rank][entries][2 words].
__device__ __forceinline__ double ringtest_value(int rank, unsigned seq, int i)
{
    return (double)(rank * 1009 + i * 17 + 1) * 1.0000001 + (double)seq * 0.25;
}
__global__ void __launch_bounds__(kBlock) k_p2p_ringtest(P2pRed pr, llword *const *rings, int entries, unsigned seq0, int rounds,
                                                         unsigned bar_seq0, unsigned long long timeout_ticks, int *status)
{
    __shared__ int s_bad, s_timeout;
    const int P = pr.nranks, me = pr.rank;
    if (threadIdx.x == 0) { s_bad = 0; s_timeout = 0; }
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const unsigned seq = seq0 + (unsigned)r;
        const size_t slot = seq % kHaloRing;
        if (r % (kHaloRing - 2) == 0) {          // flow control, as in spmv(): nobody runs more than a ring ahead
            const unsigned bs = bar_seq0 + (unsigned)(r / (kHaloRing - 2));
            for (int p = threadIdx.x; p < P; p += kBlock)
                ll_store(pr.mail[p] + mail_index(bs, P, me, kRedSlots - 1), 0.0, bs);
            for (int p = threadIdx.x; p < P; p += kBlock) {
                double v;
                if (!ll_wait(pr.mail[me] + mail_index(bs, P, p, kRedSlots - 1), bs, timeout_ticks, &v)) s_timeout = 1;
            }
            __syncthreads();
        }
        if ((r + me) & 1) __builtin_amdgcn_s_sleep(127);     // keep the ranks out of step
        for (int t = threadIdx.x; t < P * entries; t += kBlock) {
            const int p = t / entries, i = t % entries;
            ll_store(rings[p] + ((slot * P + me) * entries + i) * 2, ringtest_value(me, seq, i), seq);
        }
        for (int t = threadIdx.x; t < P * entries; t += kBlock) {
            const int p = t / entries, i = t % entries;
            double v;
            if (!ll_wait(rings[me] + ((slot * P + p) * entries + i) * 2, seq, timeout_ticks, &v)) s_timeout = 1;
            else if (!(v == ringtest_value(p, seq, i))) s_bad = 1;
        }
        __syncthreads();
        if (s_timeout) break;
    }
    if (threadIdx.x == 0) {
        if (s_bad) atomicAdd(&status[0], 1);
        if (s_timeout) atomicAdd(&status[1], 1);
    }
}

void launch_p2p_ringtest(const P2pRed &pr, llword *const *rings, int entries, unsigned seq0, int rounds, unsigned bar_seq0,
                         unsigned long long timeout_ticks, int *status, hipStream_t st)
{
    BICG_LAUNCH(k_p2p_ringtest, dim3(1), dim3(kBlock), 0, st, pr, rings, entries, seq0, rounds, bar_seq0, timeout_ticks, status);
}

void launch_p2p_selftest(const P2pRed &pr, unsigned seq0, int rounds, unsigned long long timeout_ticks, int *status,
                         hipStream_t st)
{
    BICG_LAUNCH(k_p2p_selftest, dim3(1), dim3(kBlock), 0, st, pr, seq0, rounds, timeout_ticks, status);
}

// ------------------------------------------------------------------------------------------
// fused element-wise phases
// ------------------------------------------------------------------------------------------
// Two-wide value so that one functor body serves the 16-byte vectorised main loop and the tail.
struct d2 { double a, b; };
__device__ __forceinline__ d2 operator+(d2 p, d2 q) { return {p.a + q.a, p.b + q.b}; }
__device__ __forceinline__ d2 operator-(d2 p, d2 q) { return {p.a - q.a, p.b - q.b}; }
__device__ __forceinline__ d2 operator*(d2 p, d2 q) { return {p.a * q.a, p.b * q.b}; }
__device__ __forceinline__ d2 operator*(double s, d2 q) { return {s * q.a, s * q.b}; }
__device__ __forceinline__ double hsum(d2 p) { return p.a + p.b; }
__device__ __forceinline__ double hsum(double p) { return p; }

// the same pair for launches over vectors far beyond the caches (tiled k_vec): every access of it is non-temporal
struct d2n { double a, b; };
__device__ __forceinline__ d2n operator+(d2n p, d2n q) { return {p.a + q.a, p.b + q.b}; }
__device__ __forceinline__ d2n operator-(d2n p, d2n q) { return {p.a - q.a, p.b - q.b}; }
__device__ __forceinline__ d2n operator*(d2n p, d2n q) { return {p.a * q.a, p.b * q.b}; }
__device__ __forceinline__ d2n operator*(double s, d2n q) { return {s * q.a, s * q.b}; }
__device__ __forceinline__ double hsum(d2n p) { return p.a + p.b; }

template <class T> __device__ __forceinline__ T ld(const double *p, uint32_t i);
template <> __device__ __forceinline__ double ld<double>(const double *p, uint32_t i) { return p[i]; }
template <> __device__ __forceinline__ d2n ld<d2n>(const double *p, uint32_t i)
{
    const f64x2 t = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + i));
    return {t.x, t.y};
}
template <> __device__ __forceinline__ d2 ld<d2>(const double *p, uint32_t i)
{
    const f64x2 t = *reinterpret_cast<const f64x2 *>(p + i);
    return {t.x, t.y};
}
// Streaming policy for vectors that are read and written exactly once per iteration (the solution
// x; the shifted solvers' x_j and p_j sets): non-temporal accesses keep them out of the Infinity
// Cache, which the matrix stream and the re-read work vectors use better. Measured on Transport:
// plain 150.4 -> 145.6 us, CA 170 -> 166, pipelined 168 -> 163, 16 shifts 312 -> 294 us per
// iteration. BICG_X_NT=0 / BICG_SET_NT=0 switch it off (read once).
static bool env_on_x(const char *name, bool dflt)        // measurement knob (bicg_knobs.h): the default unless built with EXPERIMENTS=1
{
    const char *v = knob_x(name);
    return v ? atoi(v) != 0 : dflt;
}
static bool stream_x() { static const bool on = env_on_x("BICG_X_NT", true); return on; }
static bool stream_sets() { static const bool on = env_on_x("BICG_SET_NT", true); return on; }

// streaming variants for vectors touched once per iteration (x): keep the Infinity Cache for the
// matrix and the vectors that are re-read soon
template <class T> __device__ __forceinline__ T ldnt(const double *p, uint32_t i);
template <> __device__ __forceinline__ double ldnt<double>(const double *p, uint32_t i) { return __builtin_nontemporal_load(p + i); }
template <> __device__ __forceinline__ d2 ldnt<d2>(const double *p, uint32_t i)
{
    const f64x2 t = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + i));
    return {t.x, t.y};
}
template <> __device__ __forceinline__ d2n ldnt<d2n>(const double *p, uint32_t i) { return ld<d2n>(p, i); }
__device__ __forceinline__ void stnt(double *p, uint32_t i, d2n v)
{
    f64x2 t; t.x = v.a; t.y = v.b;
    __builtin_nontemporal_store(t, reinterpret_cast<f64x2 *>(p + i));
}
__device__ __forceinline__ void st(double *p, uint32_t i, d2n v) { stnt(p, i, v); }
__device__ __forceinline__ void stnt(double *p, uint32_t i, double v) { __builtin_nontemporal_store(v, p + i); }
__device__ __forceinline__ void stnt(double *p, uint32_t i, d2 v)
{
    f64x2 t; t.x = v.a; t.y = v.b;
    __builtin_nontemporal_store(t, reinterpret_cast<f64x2 *>(p + i));
}
__device__ __forceinline__ void st(double *p, uint32_t i, double v) { p[i] = v; }
__device__ __forceinline__ void st(double *p, uint32_t i, d2 v)
{
    f64x2 t; t.x = v.a; t.y = v.b;
    *reinterpret_cast<f64x2 *>(p + i) = t;
}

// F::ND dot products, F::load(S) fetches the scalars once, F::apply<T>(i, acc) handles element(s) i.
// Functors of the four solvers additionally split apply into fetch (all loads of an element pair,
// none of which depends on a scalar) and compute: a kernel that finishes a dot group issues the
// loads of its first pair BEFORE waiting for the sums, so the wait runs underneath them.
// F::kModes: bit RedMode set = that instantiation is launched (keeps the others from being compiled).
template <class F, class = void> struct vec_modes { static constexpr int value = (1 << RED_TICKET) | (1 << RED_TICKET_HEAVY); };
template <class F> struct vec_modes<F, decltype((void)F::kModes)> { static constexpr int value = F::kModes; };
template <class F, class = void> struct vec_split { static constexpr bool value = false; };
template <class F> struct vec_split<F, decltype((void)F::kSplit)> { static constexpr bool value = F::kSplit; };
constexpr int kWaveOnly = 1 << RED_WAVE, kAnyMode = (1 << RED_TICKET) | (1 << RED_TICKET_HEAVY) | (1 << RED_WAVE);

// TILE > 0 (vectors far beyond the caches, vec_tiled()): a workgroup takes ONE contiguous tile of TILE pairs per thread -- 16 KiB
// of every stream for TILE = 4 --, all loads of the tile are in flight before the first store, every access is non-temporal
// (pair type d2n) and the workgroup ends: the shape that streams fastest on this GPU (bicg_stream_bench: copy 4.5 TB/s as a
// grid-stride loop, 5.9 as one workgroup per tile, 6.35 with non-temporal accesses on top). Same arithmetic per element;
// the dot partials are summed over another set of rows per workgroup than in the strided form.
constexpr int kVecTile = 4;
template <class F, int MODE, int TILE = 0>
__global__ void __launch_bounds__(kBlock) k_vec(F f, uint32_t n, Scal *S, Reduce red, Finish fin)
{
    constexpr int ND = F::ND > 0 ? F::ND : 1;
    double acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = 0.0;
    const uint32_t npair = n >> 1;
    const uint32_t i0 = blockIdx.x * kBlock + threadIdx.x, stride = gridDim.x * kBlock;
    if constexpr (TILE > 0) {
        static_assert(vec_split<F>::value, "tiled launches need the functor's fetch / compute split");
        const uint32_t t0 = blockIdx.x * (uint32_t)(kBlock * TILE) + threadIdx.x;
        typename F::template In<d2n> in[TILE];
        auto fetch_tile = [&]() {
#pragma unroll
            for (int u = 0; u < TILE; ++u) {
                const uint32_t i = t0 + (uint32_t)u * kBlock;
                if (i < npair) in[u] = f.template fetch<d2n>(2 * i);
            }
        };
        auto compute_tile = [&]() {
#pragma unroll
            for (int u = 0; u < TILE; ++u) {
                const uint32_t i = t0 + (uint32_t)u * kBlock;
                if (i < npair) f.template compute<d2n>(2 * i, in[u], acc);
            }
            if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) f.template apply<double>(n - 1, acc);
        };
        if constexpr (MODE == RED_WAVE) {
            __shared__ FinishLds fl;
            __shared__ Scal priv;
            const Scal *sc = S;
            const bool helper = fin.seq && (fin.roles & FIN_SHARDS) && blockIdx.x < (unsigned)kShards;      // (see below)
            if (!helper) fetch_tile();
            if (fin.seq) sc = finish_group(S, fin, fin.roles, blockIdx.x, gridDim.x, fl, &priv);
            if (helper) fetch_tile();
            if (sc->done) return;
            f.load(sc);
            compute_tile();
            if (F::ND > 0) wave_publish<ND>(acc, red.partial, red.slot_base + blockIdx.x);
        } else {
            if (S->done) return;
            __shared__ double sm[5 * ND];
            f.load(S);
            fetch_tile();
            compute_tile();
            if (F::ND > 0) reduce_publish<ND, MODE == RED_TICKET_HEAVY>(acc, S, red, blockIdx.x, sm);
        }
        return;
    }
    if constexpr (MODE == RED_WAVE) {
        __shared__ FinishLds fl;
        __shared__ Scal priv;
        const Scal *sc = S;
        if constexpr (vec_split<F>::value) {
            typename F::template In<d2> pre{};
            const bool have = i0 < npair;
            // Loads return in issue order (vmcnt), so a workgroup that sums a shard must not queue its
            // partial-sum loads behind its own vector loads: the shard totals are what every other
            // workgroup of the launch is waiting for. Everybody else fetches first and waits underneath.
            const bool helper = fin.seq && (fin.roles & FIN_SHARDS) && blockIdx.x < (unsigned)kShards;
            if (have && !helper) pre = f.template fetch<d2>(2 * i0);
            if (fin.seq) sc = finish_group(S, fin, fin.roles, blockIdx.x, gridDim.x, fl, &priv);
            if (have && helper) pre = f.template fetch<d2>(2 * i0);
            if (sc->done) return;
            f.load(sc);
            if (have) f.template compute<d2>(2 * i0, pre, acc);
            for (uint32_t i = i0 + stride; i < npair; i += stride) f.template compute<d2>(2 * i, f.template fetch<d2>(2 * i), acc);
        } else {
            if (fin.seq) sc = finish_group(S, fin, fin.roles, blockIdx.x, gridDim.x, fl, &priv);
            if (sc->done) return;
            f.load(sc);
            for (uint32_t i = i0; i < npair; i += stride) f.template apply<d2>(2 * i, acc);
        }
        if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) f.template apply<double>(n - 1, acc);
        if (F::ND > 0) wave_publish<ND>(acc, red.partial, red.slot_base + blockIdx.x);
    } else {
        if (S->done) return;
        __shared__ double sm[5 * ND];
        f.load(S);
        for (uint32_t i = i0; i < npair; i += stride) f.template apply<d2>(2 * i, acc);
        if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) f.template apply<double>(n - 1, acc);
        if (F::ND > 0) reduce_publish<ND, MODE == RED_TICKET_HEAVY>(acc, S, red, blockIdx.x, sm);
    }
}

static unsigned g_vec_grid_cap = 0;
void set_vec_grid_cap(unsigned cap) { g_vec_grid_cap = cap; }
// pairs per thread of an element-wise launch over n rows (0: grid-stride loop over <= kMaxGrid workgroups)
static unsigned vec_ppt(uint32_t n)
{
    static const int ppt_env = [] { const char *v = knob_x("BICG_VEC_PPT"); return v ? atoi(v) : -1; }();
    // (tiles from 2^24 rows = 128 MiB per vector: 256^3 plain 0.512 -> 0.442, CA 0.662 -> 0.573 ms per iteration; at 6.4 M rows --
    // 49 MiB per vector, what one kernel writes the next still finds in the Infinity Cache -- plain loses 3.6 %, pipelined gains 1.7 %;
    // at Transport size plain loses 7 %: profiles/r05/ab_vec_tile_sizes.txt)
    return ppt_env >= 0 ? (unsigned)ppt_env : (n >= (1u << 24) ? (unsigned)kVecTile : 0u);
}
// ... as contiguous tiles with non-temporal accesses (k_vec<.., TILE>; BICG_VEC_TILE=0: the strided form of round 4)
static bool vec_tiled(uint32_t n)
{
    static const bool on = env_on_x("BICG_VEC_TILE", true);
    return on && !g_vec_grid_cap && vec_ppt(n) == (unsigned)kVecTile;
}
unsigned vec_grid(uint32_t n)
{
    // 256 CUs x 8 resident workgroups, grid-stride beyond (BICG_VEC_GRID: measurement knob, <= kMaxGrid)
    static const unsigned env_cap = [] {
        const char *v = knob_x("BICG_VEC_GRID");
        const int g = v ? atoi(v) : kMaxGrid;
        return (unsigned)(g >= 1 && g <= kMaxGrid ? g : kMaxGrid);
    }();
    // Vectors far beyond the caches (>= 16.8 M rows): one workgroup per tile of 4 element pairs per thread (16 KiB per stream)
    // instead of <= 2048 persistent workgroups striding through the vectors -- short workgroups stream faster (STREAM read:
    // +12 %, profiles/NOTES.md; 512^3 Laplacian: plain 9.12 -> 8.65 ms, CA 10.86 -> 10.10 ms per iteration, round 4). At
    // 16.8 M rows (256^3) and at Transport size the two forms tie. BICG_VEC_PPT=p forces p pairs per thread everywhere
    // (0: never); the grid is capped at the partial-sum slots every context has.
    unsigned g = ((n >> 1) + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    const unsigned ppt = vec_ppt(n);
    if (ppt && !g_vec_grid_cap) {
        g = (g + ppt - 1) / ppt;
        const unsigned slots = std::max<unsigned>(kMaxGrid, (n + kGroupRows - 1) / kGroupRows);      // ctx_state: nslots >= row groups
        return g < slots ? g : slots;
    }
    const unsigned cap = g_vec_grid_cap && g_vec_grid_cap < env_cap ? g_vec_grid_cap : env_cap;
    if (g > cap) g = cap;
    return g;
}

template <class F>
static void run_vec(F f, uint32_t n, const Launch &L, Reduce red)
{
    const unsigned g = vec_grid(n);
    red.expected = g;
    red.slot_base = 0;
    constexpr int modes = vec_modes<F>::value;
    const int mode = modes == kWaveOnly ? RED_WAVE : red_mode(red, L.fin, F::ND > 0);
    if (!((modes >> mode) & 1)) {
        fprintf(stderr, "ERROR: bicgstab_hip: element-wise kernel launched in reduction mode %d it is not built for\n", mode);
        abort();
    }
    if constexpr (vec_split<F>::value) {
        if (vec_tiled(n)) {
            if constexpr ((modes >> RED_WAVE) & 1)
                if (mode == RED_WAVE) { BICG_LAUNCH((k_vec<F, RED_WAVE, kVecTile>), dim3(g), dim3(kBlock), 0, L.st, f, n, L.S, red, L.fin); return; }
            if constexpr (((modes >> RED_TICKET_HEAVY) & 1) && F::ND > 0)
                if (mode == RED_TICKET_HEAVY) { BICG_LAUNCH((k_vec<F, RED_TICKET_HEAVY, kVecTile>), dim3(g), dim3(kBlock), 0, L.st, f, n, L.S, red, L.fin); return; }
            if constexpr ((modes >> RED_TICKET) & 1)
                if (mode != RED_WAVE) { BICG_LAUNCH((k_vec<F, RED_TICKET, kVecTile>), dim3(g), dim3(kBlock), 0, L.st, f, n, L.S, red, L.fin); return; }
        }
    }
    if constexpr ((modes >> RED_WAVE) & 1)
        if (mode == RED_WAVE) { BICG_LAUNCH((k_vec<F, RED_WAVE>), dim3(g), dim3(kBlock), 0, L.st, f, n, L.S, red, L.fin); return; }
    if constexpr (((modes >> RED_TICKET_HEAVY) & 1) && F::ND > 0)
        if (mode == RED_TICKET_HEAVY) { BICG_LAUNCH((k_vec<F, RED_TICKET_HEAVY>), dim3(g), dim3(kBlock), 0, L.st, f, n, L.S, red, L.fin); return; }
    if constexpr ((modes >> RED_TICKET) & 1)
        BICG_LAUNCH((k_vec<F, RED_TICKET>), dim3(g), dim3(kBlock), 0, L.st, f, n, L.S, red, L.fin);
}
// shifted solvers and kernel-level entry points: scalar block updated in place, ticket reductions
template <class F>
static void run_vec(F f, uint32_t n, Scal *S, Reduce red, hipStream_t stream)
{
    run_vec(f, n, Launch{S, Finish{}, stream}, red);
}

// ---- init: r = b - Ax ; r# = r ; [p = r] ; [bsave = b] ; (r,r)     (src/solver.c:74-78, 475-479)
struct FInit {
    static constexpr int ND = 1;
    static constexpr int kModes = kAnyMode;
    double *r, *rh, *p, *bs; const double *ax;
    __device__ void load(const Scal *) {}
    template <class T> __device__ void apply(uint32_t i, double *acc) const
    {
        T b = ld<T>(r, i);
        if (bs) st(bs, i, b);
        T rr = b + (-1.0) * ld<T>(ax, i);
        st(r, i, rr); st(rh, i, rr);
        if (p) st(p, i, rr);
        acc[0] += hsum(rr * rr);
    }
};
void launch_init_residual(const Vecs &v, bool copy_p, bool save_b, const Launch &L, Reduce red)
{
    run_vec(FInit{v.r, v.rh, copy_p ? v.p : nullptr, save_b ? v.b : nullptr, v.ax}, v.n, L, red);
}

// ---- plain: q = r - alpha s (kept in r)                               (src/solver.c:94)
struct FPlainQ {
    static constexpr int ND = 0;
    static constexpr int kModes = kAnyMode;
    static constexpr bool kSplit = true;
    double *r; const double *s; double alpha;
    template <class T> struct In { T r, s; };
    __device__ void load(const Scal *S) { alpha = S->alpha; }
    template <class T> __device__ In<T> fetch(uint32_t i) const { return {ld<T>(r, i), ld<T>(s, i)}; }
    template <class T> __device__ void compute(uint32_t i, const In<T> &in, double *) const
    {
        st(r, i, in.r + (-alpha) * in.s);
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const { compute<T>(i, fetch<T>(i), acc); }
};
void launch_plain_q(const Vecs &v, const Launch &L) { run_vec(FPlainQ{v.r, v.s, 0.0}, v.n, L, Reduce{}); }

// ---- plain: x += alpha p + omega q ; r = q - omega y ; (r,r), (r#,r)   (src/solver.c:105-111)
template <bool XNT> struct FPlainXR {
    static constexpr int ND = 2;
    static constexpr int kModes = kAnyMode;
    static constexpr bool kSplit = true;
    double *x, *r; const double *q, *p, *y, *rh; double alpha, omega;      // q: where q lives (r itself, or the fused iteration's own buffer)
    template <class T> struct In { T q, x, p, y, rh; };
    __device__ void load(const Scal *S) { alpha = S->alpha; omega = S->omega; }
    template <class T> __device__ In<T> fetch(uint32_t i) const
    {
        return {ld<T>(q, i), XNT ? ldnt<T>(x, i) : ld<T>(x, i), ld<T>(p, i), ld<T>(y, i), ld<T>(rh, i)};
    }
    template <class T> __device__ void compute(uint32_t i, const In<T> &in, double *acc) const
    {
        T xx = in.x + alpha * in.p;
        xx = xx + omega * in.q;
        if (XNT) stnt(x, i, xx); else st(x, i, xx);
        T rr = in.q + (-omega) * in.y;
        st(r, i, rr);
        acc[0] += hsum(rr * rr);
        acc[1] += hsum(in.rh * rr);
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const { compute<T>(i, fetch<T>(i), acc); }
};
void launch_plain_xr(const Vecs &v, const Launch &L, Reduce red, const double *q)
{
    if (stream_x()) run_vec(FPlainXR<true>{v.x, v.r, q ? q : v.r, v.p, v.y, v.rh, 0.0, 0.0}, v.n, L, red);
    else run_vec(FPlainXR<false>{v.x, v.r, q ? q : v.r, v.p, v.y, v.rh, 0.0, 0.0}, v.n, L, red);
}

// ---- plain: p = beta p ; p += r ; p += (-beta*omega) s                (src/solver.c:117-119)
struct FPlainP {
    static constexpr int ND = 0;
    static constexpr int kModes = kAnyMode;
    static constexpr bool kSplit = true;
    double *p; const double *r, *s; double beta, c;
    template <class T> struct In { T p, r, s; };
    __device__ void load(const Scal *S) { beta = S->beta; c = -S->beta * S->omega; }
    template <class T> __device__ In<T> fetch(uint32_t i) const { return {ld<T>(p, i), ld<T>(r, i), ld<T>(s, i)}; }
    template <class T> __device__ void compute(uint32_t i, const In<T> &in, double *) const
    {
        T pp = beta * in.p;
        pp = pp + 1.0 * in.r;
        pp = pp + c * in.s;
        st(p, i, pp);
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const { compute<T>(i, fetch<T>(i), acc); }
};
void launch_plain_p(const Vecs &v, const Launch &L) { run_vec(FPlainP{v.p, v.r, v.s, 0.0, 0.0}, v.n, L, Reduce{}); }

// ---- CA: p = r + beta(p - omega s) ; s = w + beta(s - omega z)         (src/solver.c:217-222)
struct FCaPS {
    static constexpr int ND = 0;
    static constexpr int kModes = kAnyMode;
    static constexpr bool kSplit = true;
    double *p, *s; const double *r, *z, *w; double beta, omega;
    template <class T> struct In { T p, s, r, z, w; };
    __device__ void load(const Scal *S) { beta = S->beta; omega = S->omega; }
    template <class T> __device__ In<T> fetch(uint32_t i) const
    {
        return {ld<T>(p, i), ld<T>(s, i), ld<T>(r, i), ld<T>(z, i), ld<T>(w, i)};
    }
    template <class T> __device__ void compute(uint32_t i, const In<T> &in, double *) const
    {
        st(p, i, recur3<T>(in.p, in.s, in.r, omega, beta));
        st(s, i, recur3<T>(in.s, in.z, in.w, omega, beta));
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const { compute<T>(i, fetch<T>(i), acc); }
};
void launch_ca_ps(const Vecs &v, const Launch &L) { run_vec(FCaPS{v.p, v.s, v.r, v.z, v.w, 0.0, 0.0}, v.n, L, Reduce{}); }

// ---- q = r - alpha s (in r) ; y = w - alpha z (in w) ; (q,y), (y,y)    (src/solver.c:225-228, 361-364)
struct FQY {
    static constexpr int ND = 2;
    static constexpr int kModes = kAnyMode;
    static constexpr bool kSplit = true;
    double *r, *w; const double *s, *z; double alpha;
    template <class T> struct In { T r, s, w, z; };
    __device__ void load(const Scal *S) { alpha = S->alpha; }
    template <class T> __device__ In<T> fetch(uint32_t i) const { return {ld<T>(r, i), ld<T>(s, i), ld<T>(w, i), ld<T>(z, i)}; }
    template <class T> __device__ void compute(uint32_t i, const In<T> &in, double *acc) const
    {
        T q = in.r + (-alpha) * in.s;
        T y = in.w + (-alpha) * in.z;
        st(r, i, q); st(w, i, y);
        acc[0] += hsum(q * y);
        acc[1] += hsum(y * y);
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const { compute<T>(i, fetch<T>(i), acc); }
};
void launch_qy(const Vecs &v, const Launch &L, Reduce red) { run_vec(FQY{v.r, v.w, v.s, v.z, 0.0}, v.n, L, red); }

// ---- CA: x += alpha p + omega q ; r = q - omega y ; (r,r), (r#,r), [slot 2 left for (r#,w)], (r#,s), (r#,z)
//      (src/solver.c:233-236, 240, 242-243; (r#,w) comes from the following SpMV's epilogue)
template <bool XNT> struct FCaXR {
    static constexpr int ND = 5;
    static constexpr int kModes = kAnyMode;
    static constexpr bool kSplit = true;
    double *x, *r; const double *p, *w, *rh, *s, *z; double alpha, omega;
    template <class T> struct In { T q, x, p, w, rh, s, z; };
    __device__ void load(const Scal *S) { alpha = S->alpha; omega = S->omega; }
    template <class T> __device__ In<T> fetch(uint32_t i) const
    {
        return {ld<T>(r, i), XNT ? ldnt<T>(x, i) : ld<T>(x, i), ld<T>(p, i), ld<T>(w, i), ld<T>(rh, i), ld<T>(s, i), ld<T>(z, i)};
    }
    template <class T> __device__ void compute(uint32_t i, const In<T> &in, double *acc) const
    {
        T xx = in.x + alpha * in.p;
        xx = xx + omega * in.q;
        if (XNT) stnt(x, i, xx); else st(x, i, xx);
        T rr = in.q + (-omega) * in.w;
        st(r, i, rr);
        acc[0] += hsum(rr * rr);
        acc[1] += hsum(in.rh * rr);
        acc[3] += hsum(in.rh * in.s);
        acc[4] += hsum(in.rh * in.z);
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const { compute<T>(i, fetch<T>(i), acc); }
};
void launch_ca_xr(const Vecs &v, const Launch &L, Reduce red)
{
    red.p2p.mask &= ~4u;    // slot 2 belongs to the following SpMV's epilogue
    if (stream_x()) run_vec(FCaXR<true>{v.x, v.r, v.p, v.w, v.rh, v.s, v.z, 0.0, 0.0}, v.n, L, red);
    else run_vec(FCaXR<false>{v.x, v.r, v.p, v.w, v.rh, v.s, v.z, 0.0, 0.0}, v.n, L, red);
}

// ---- pipelined phase 1: p, s, z recurrences ; q, y ; (q,y), (y,y)       (src/solver.c:352-364)
// y = w - alpha z goes to its own vector yb (the reference keeps it in w, :363-364): w is the input of the
// SpMV t = A w in whose epilogue this phase may run (k_spmv_sell_epi), so it cannot be overwritten there
struct FPipe1 {
    static constexpr int ND = 2;
    static constexpr int kModes = kWaveOnly;
    static constexpr bool kSplit = true;
    double *p, *s, *z, *r, *yb; const double *w, *t, *v; double alpha, beta, omega;
    template <class T> struct In { T r, w, s, z, p, v, t; };
    __device__ void load(const Scal *S) { alpha = S->alpha; beta = S->beta; omega = S->omega; }
    template <class T> __device__ In<T> fetch(uint32_t i) const
    {
        return {ld<T>(r, i), ld<T>(w, i), ld<T>(s, i), ld<T>(z, i), ld<T>(p, i), ld<T>(v, i), ld<T>(t, i)};
    }
    template <class T> __device__ void compute(uint32_t i, const In<T> &in, double *acc) const
    {
        st(p, i, recur3<T>(in.p, in.s, in.r, omega, beta));
        T s1 = recur3<T>(in.s, in.z, in.w, omega, beta);
        T z1 = recur3<T>(in.z, in.v, in.t, omega, beta);
        st(s, i, s1); st(z, i, z1);
        T q = in.r + (-alpha) * s1;
        T y = in.w + (-alpha) * z1;
        st(r, i, q); st(yb, i, y);
        acc[0] += hsum(q * y);
        acc[1] += hsum(y * y);
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const { compute<T>(i, fetch<T>(i), acc); }
};
void launch_pipe_f1(const Vecs &v, const Launch &L, Reduce red)
{
    run_vec(FPipe1{v.p, v.s, v.z, v.r, v.y, v.w, v.t, v.v, 0.0, 0.0, 0.0}, v.n, L, red);
}

// ---- pipelined phase 2: x ; r = q - omega y ; w = y - omega (t - alpha v) ; five dots  (src/solver.c:370-380)
// t - alpha v is not written back: t is overwritten by the next SpMV (src/solver.c:381).
template <bool XNT> struct FPipe2 {
    static constexpr int ND = 5;
    static constexpr int kModes = kWaveOnly;
    static constexpr bool kSplit = true;
    double *x, *r, *w; const double *yb, *p, *t, *v, *rh, *s, *z; double alpha, omega;
    template <class T> struct In { T q, y, x, p, t, v, rh, s, z; };
    __device__ void load(const Scal *S) { alpha = S->alpha; omega = S->omega; }
    template <class T> __device__ In<T> fetch(uint32_t i) const
    {
        return {ld<T>(r, i), ld<T>(yb, i), XNT ? ldnt<T>(x, i) : ld<T>(x, i), ld<T>(p, i), ld<T>(t, i), ld<T>(v, i), ld<T>(rh, i),
                ld<T>(s, i), ld<T>(z, i)};
    }
    template <class T> __device__ void compute(uint32_t i, const In<T> &in, double *acc) const
    {
        T xx = in.x + alpha * in.p;
        xx = xx + omega * in.q;
        if (XNT) stnt(x, i, xx); else st(x, i, xx);
        T rr = in.q + (-omega) * in.y;
        st(r, i, rr);
        T tt = in.t + (-alpha) * in.v;
        T ww = in.y + (-omega) * tt;
        st(w, i, ww);
        acc[0] += hsum(rr * rr);
        acc[1] += hsum(in.rh * rr);
        acc[2] += hsum(in.rh * ww);
        acc[3] += hsum(in.rh * in.s);
        acc[4] += hsum(in.rh * in.z);
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const { compute<T>(i, fetch<T>(i), acc); }
};
void launch_pipe_f2(const Vecs &v, const Launch &L, Reduce red)
{
    if (stream_x()) run_vec(FPipe2<true>{v.x, v.r, v.w, v.y, v.p, v.t, v.v, v.rh, v.s, v.z, 0.0, 0.0}, v.n, L, red);
    else run_vec(FPipe2<false>{v.x, v.r, v.w, v.y, v.p, v.t, v.v, v.rh, v.s, v.z, 0.0, 0.0}, v.n, L, red);
}

// ---- residual replacement pieces
struct FPUpdate {   // p = r + beta (p - omega s)                            (src/solver.c:494-496)
    static constexpr int ND = 0;
    static constexpr int kModes = kWaveOnly;
    double *p; const double *s, *r; double beta, omega;
    __device__ void load(const Scal *S) { beta = S->beta; omega = S->omega; }
    template <class T> __device__ void apply(uint32_t i, double *) const
    {
        st(p, i, recur3<T>(ld<T>(p, i), ld<T>(s, i), ld<T>(r, i), omega, beta));
    }
};
void launch_p_update(const Vecs &v, const Launch &L) { run_vec(FPUpdate{v.p, v.s, v.r, 0.0, 0.0}, v.n, L, Reduce{}); }

struct FXUpdate {   // x += alpha p ; x += omega q                           (src/solver.c:519-520)
    static constexpr int ND = 0;
    static constexpr int kModes = kWaveOnly;
    double *x; const double *p, *r; double alpha, omega;
    __device__ void load(const Scal *S) { alpha = S->alpha; omega = S->omega; }
    template <class T> __device__ void apply(uint32_t i, double *) const
    {
        T xx = ld<T>(x, i) + alpha * ld<T>(p, i);
        st(x, i, xx + omega * ld<T>(r, i));
    }
};
void launch_x_update(const Vecs &v, const Launch &L) { run_vec(FXUpdate{v.x, v.p, v.r, 0.0, 0.0}, v.n, L, Reduce{}); }

struct FTrueRes {   // r = b ; r += -1.0 * Ax                                (src/solver.c:524-525)
    static constexpr int ND = 0;
    static constexpr int kModes = kWaveOnly;
    double *r; const double *b, *ax;
    __device__ void load(const Scal *) {}
    template <class T> __device__ void apply(uint32_t i, double *) const
    {
        st(r, i, ld<T>(b, i) + (-1.0) * ld<T>(ax, i));
    }
};
void launch_true_residual(const Vecs &v, const Launch &L) { run_vec(FTrueRes{v.r, v.b, v.ax}, v.n, L, Reduce{}); }

struct FDots5 {     // (r,r), (r#,r), (r#,w), (r#,s), (r#,z)                 (src/solver.c:533-538)
    static constexpr int ND = 5;
    static constexpr int kModes = kWaveOnly;
    const double *r, *rh, *w, *s, *z;
    __device__ void load(const Scal *) {}
    template <class T> __device__ void apply(uint32_t i, double *acc) const
    {
        T rr = ld<T>(r, i), h = ld<T>(rh, i);
        acc[0] += hsum(rr * rr);
        acc[1] += hsum(h * rr);
        acc[2] += hsum(h * ld<T>(w, i));
        acc[3] += hsum(h * ld<T>(s, i));
        acc[4] += hsum(h * ld<T>(z, i));
    }
};
void launch_dots5(const Vecs &v, const Launch &L, Reduce red) { run_vec(FDots5{v.r, v.rh, v.w, v.s, v.z}, v.n, L, red); }

// ---- shifted BiCGStab (reference src/shifted_solver.c:182-354) ---------------------------------
struct FShiftInit {   // r# = r ; p[seed] = r ; (r,r)                                  (:238-250)
    static constexpr int ND = 1;
    const double *r; double *rh, *ps;
    __device__ void load(const Scal *) {}
    template <class T> __device__ void apply(uint32_t i, double *acc) const
    {
        T rr = ld<T>(r, i);
        st(rh, i, rr); st(ps, i, rr);
        acc[0] += hsum(rr * rr);
    }
};
void launch_shift_init(const Vecs &v, double *p_seed, Scal *S, Reduce red, hipStream_t s)
{
    run_vec(FShiftInit{v.r, v.rh, p_seed}, v.n, S, red, s);
}

struct FShiftQ {      // r_old = r ; q = r - alpha[seed] s (kept in r)                  (:269, 275)
    static constexpr int ND = 0;
    double *r, *rold; const double *s; double alpha;
    __device__ void load(const Scal *S) { alpha = S->alpha; }
    template <class T> __device__ void apply(uint32_t i, double *) const
    {
        T r0 = ld<T>(r, i);
        st(rold, i, r0);
        st(r, i, r0 + (-alpha) * ld<T>(s, i));
    }
};
void launch_shift_q(const Vecs &v, Scal *S, hipStream_t s) { run_vec(FShiftQ{v.r, v.ax, v.s, 0.0}, v.n, S, Reduce{}, s); }

// One pass over BOTH vector sets: the seed's x and r, the two dots, and for every other shift j the
// p_j update that the reference does at the top of the iteration (:264-266, with r = r_old), the
// x_j update (:296-297) and the second p_j update (:298-299) -- the same operations on every
// element in the same order, but p_j and x_j are read and written ONCE per iteration
// (32 n bytes per shift instead of the reference's 136 n).
template <bool SNT> struct FShiftUpdate {
    static constexpr int ND = 2;
    double *xs, *r, *pset, *xset; const double *ps, *y, *rh, *rold; const ShiftDev *H; uint32_t stride;
    double alpha, omega; int nsig, seed;
    const double *beta_j, *alpha_j, *cp, *cx, *c1, *c2;
    __device__ void load(const Scal *S)
    {
        alpha = S->alpha; omega = S->omega;
        nsig = H->nsig; seed = H->seed;
        beta_j = H->beta; alpha_j = H->alpha; cp = H->cp; cx = H->cx; c1 = H->c1; c2 = H->c2;
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const
    {
        const T q = ld<T>(r, i), ro = ld<T>(rold, i);
        T xx = ld<T>(xs, i) + alpha * ld<T>(ps, i);             // x[seed] += alpha p[seed] ; += omega q   (:292-293)
        st(xs, i, xx + omega * q);
        for (int j = 0; j < nsig; ++j) {
            if (j == seed) continue;
            double *pj = pset + (size_t)j * stride, *xj = xset + (size_t)j * stride;
            T p = beta_j[j] * (SNT ? ldnt<T>(pj, i) : ld<T>(pj, i));   // my_dscal(beta[j])                 (:265)
            p = p + cp[j] * ro;                                  // += 1/(pi zeta) r   (r == r_old here)    (:266)
            T x = (SNT ? ldnt<T>(xj, i) : ld<T>(xj, i)) + cx[j] * q;   // (:296)
            x = x + alpha_j[j] * p;                              // (:297)
            if (SNT) stnt(xj, i, x); else st(xj, i, x);
            p = p + c1[j] * q;                                   // (:298)
            p = p + c2[j] * ro;                                  // (:299)
            if (SNT) stnt(pj, i, p); else st(pj, i, p);
        }
        const T rr = q + (-omega) * ld<T>(y, i);                // r = q - omega y                         (:303)
        st(r, i, rr);
        acc[0] += hsum(rr * rr);
        acc[1] += hsum(ld<T>(rh, i) * rr);
    }
};
void launch_shift_update(const Vecs &v, double *p_set, double *x_set, uint32_t set_stride, int seed, const ShiftDev *H,
                         Scal *S, Reduce red, hipStream_t s)
{
    auto go = [&](auto f) {
        f.xs = x_set + (size_t)seed * set_stride; f.ps = p_set + (size_t)seed * set_stride;
        f.r = v.r; f.pset = p_set; f.xset = x_set; f.y = v.y; f.rh = v.rh; f.rold = v.ax; f.H = H; f.stride = set_stride;
        run_vec(f, v.n, S, red, s);
    };
    if (stream_sets()) go(FShiftUpdate<true>{}); else go(FShiftUpdate<false>{});
}

// ---- seed-switching shifted solvers (reference src/shifted_switching_solver.c:376-446) -----------
struct FSwQ {         // r_old = r ; q = r - alpha s -> r and q_copy                       (:376, 393-394)
    static constexpr int ND = 0;
    double *r, *rold, *qc; const double *s; double alpha;
    __device__ void load(const Scal *S) { alpha = S->alpha; }
    template <class T> __device__ void apply(uint32_t i, double *) const
    {
        T r0 = ld<T>(r, i);
        st(rold, i, r0);
        T q = r0 + (-alpha) * ld<T>(s, i);
        st(r, i, q); st(qc, i, q);
    }
};
void launch_sw_q(const Vecs &v, double *qcopy, Scal *S, hipStream_t s) { run_vec(FSwQ{v.r, v.ax, qcopy, v.s, 0.0}, v.n, S, Reduce{}, s); }

struct FSwSeed {      // x[seed] += alpha p[seed] ; += omega q ; r = q - omega y ; (r,r), (r#,r)   (:413-418)
    static constexpr int ND = 2;
    double *xs, *r; const double *ps, *y, *rh; double alpha, omega;
    __device__ void load(const Scal *S) { alpha = S->alpha; omega = S->omega; }
    template <class T> __device__ void apply(uint32_t i, double *acc) const
    {
        const T q = ld<T>(r, i);
        T xx = ld<T>(xs, i) + alpha * ld<T>(ps, i);
        st(xs, i, xx + omega * q);
        const T rr = q + (-omega) * ld<T>(y, i);
        st(r, i, rr);
        acc[0] += hsum(rr * rr);
        acc[1] += hsum(ld<T>(rh, i) * rr);
    }
};
void launch_sw_seed(const Vecs &v, double *x_seed, const double *p_seed, Scal *S, Reduce red, hipStream_t s)
{
    run_vec(FSwSeed{x_seed, v.r, p_seed, v.y, v.rh, 0.0, 0.0}, v.n, S, red, s);
}

// p[seed] = beta p[seed] + r - beta omega s (:423-425) and, for every shift that is neither the seed
// nor frozen, the six updates of :438-446 in the reference's order; each p_j and x_j is read and
// written once.
template <bool SNT> struct FSwShifts {
    static constexpr int ND = 0;
    double *ps, *pset, *xset; const double *r, *s, *qc, *rold; const ShiftDev *H; uint32_t stride;
    double beta, omega; int nsig;
    const int *skip; const double *beta_j, *alpha_j, *cp, *cx, *c1, *c2;
    __device__ void load(const Scal *S)
    {
        beta = S->beta; omega = S->omega; nsig = H->nsig;
        skip = H->skip; beta_j = H->beta; alpha_j = H->alpha; cp = H->cp; cx = H->cx; c1 = H->c1; c2 = H->c2;
    }
    template <class T> __device__ void apply(uint32_t i, double *) const
    {
        const T rr = ld<T>(r, i), q = ld<T>(qc, i), ro = ld<T>(rold, i);
        T p = beta * ld<T>(ps, i);                               // my_dscal(beta)            (:423)
        p = p + 1.0 * rr;                                        // += 1.0 r                  (:424)
        p = p + (-beta * omega) * ld<T>(s, i);                   // += (-beta omega) s        (:425)
        st(ps, i, p);
        for (int j = 0; j < nsig; ++j) {
            if (skip[j]) continue;
            double *pj = pset + (size_t)j * stride, *xj = xset + (size_t)j * stride;
            T pp = SNT ? ldnt<T>(pj, i) : ld<T>(pj, i);
            T x = (SNT ? ldnt<T>(xj, i) : ld<T>(xj, i)) + cx[j] * q;     // (:438)
            x = x + alpha_j[j] * pp;                                 // (:439)
            if (SNT) stnt(xj, i, x); else st(xj, i, x);
            pp = pp + c1[j] * q;                                     // (:440)
            pp = pp + c2[j] * ro;                                    // (:441)
            pp = beta_j[j] * pp;                                     // my_dscal(beta_j)      (:444)
            pp = pp + cp[j] * rr;                                    // (:445)
            if (SNT) stnt(pj, i, pp); else st(pj, i, pp);
        }
    }
};
void launch_sw_shifts(const Vecs &v, const double *qcopy, double *p_set, double *x_set, uint32_t set_stride, int seed,
                      const ShiftDev *H, Scal *S, hipStream_t s)
{
    auto go = [&](auto f) {
        f.ps = p_set + (size_t)seed * set_stride; f.pset = p_set; f.xset = x_set; f.r = v.r; f.s = v.s; f.qc = qcopy;
        f.rold = v.ax; f.H = H; f.stride = set_stride;
        run_vec(f, v.n, S, Reduce{}, s);
    };
    if (stream_sets()) go(FSwShifts<true>{}); else go(FSwShifts<false>{});
}

// x <- a x (my_dscal, src/vector.c:17-21); runs while the device is paused at a seed switch
__global__ void __launch_bounds__(kBlock) k_scale(double *x, uint32_t n, double a)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) x[i] = a * x[i];
}
void launch_scale(double *x, uint32_t n, double a, hipStream_t s)
{
    if (n == 0) return;
    unsigned g = (n + kBlock - 1) / kBlock;
    if (g > (unsigned)kMaxGrid) g = kMaxGrid;
    BICG_LAUNCH(k_scale, dim3(g), dim3(kBlock), 0, s, x, n, a);
}

// ---- pipelined shifted variant (reference src/shifted_solver.c:794-843)
struct FShPipe1 {   // p[seed], s, z recurrences ; r_old = r ; q, y ; (q,y), (y,y)          (:794-813)
    static constexpr int ND = 2;
    double *p, *s, *z, *r, *w, *rold; const double *t, *v; double alpha, beta, omega;
    __device__ void load(const Scal *S) { alpha = S->alpha; beta = S->beta; omega = S->omega; }
    template <class T> __device__ void apply(uint32_t i, double *acc) const
    {
        T r0 = ld<T>(r, i), w0 = ld<T>(w, i), s0 = ld<T>(s, i), z0 = ld<T>(z, i);
        st(p, i, recur3<T>(ld<T>(p, i), s0, r0, omega, beta));
        T s1 = recur3<T>(s0, z0, w0, omega, beta);
        T z1 = recur3<T>(z0, ld<T>(v, i), ld<T>(t, i), omega, beta);
        st(s, i, s1); st(z, i, z1);
        st(rold, i, r0);
        T q = r0 + (-alpha) * s1;
        T y = w0 + (-alpha) * z1;
        st(r, i, q); st(w, i, y);
        acc[0] += hsum(q * y);
        acc[1] += hsum(y * y);
    }
};
void launch_shift_pipe1(const Vecs &v, double *p_seed, Scal *S, Reduce red, hipStream_t s)
{
    run_vec(FShPipe1{p_seed, v.s, v.z, v.r, v.w, v.ax, v.t, v.v, 0.0, 0.0, 0.0}, v.n, S, red, s);
}

template <bool SNT> struct FShPipe2 {   // x[seed] ; every p_j, x_j ; r ; w = y - omega (t - alpha v) ; five dots   (:829-848)
    static constexpr int ND = 5;
    double *xs, *r, *w, *pset, *xset; const double *ps, *t, *v, *rh, *s, *z, *rold; const ShiftDev *H; uint32_t stride;
    double alpha, omega; int nsig, seed;
    const double *beta_j, *alpha_j, *cp, *cx, *c1, *c2;
    __device__ void load(const Scal *S)
    {
        alpha = S->alpha; omega = S->omega;
        nsig = H->nsig; seed = H->seed;
        beta_j = H->beta; alpha_j = H->alpha; cp = H->cp; cx = H->cx; c1 = H->c1; c2 = H->c2;
    }
    template <class T> __device__ void apply(uint32_t i, double *acc) const
    {
        const T q = ld<T>(r, i), y = ld<T>(w, i), ro = ld<T>(rold, i);
        T xx = ld<T>(xs, i) + alpha * ld<T>(ps, i);
        st(xs, i, xx + omega * q);
        for (int j = 0; j < nsig; ++j) {
            if (j == seed) continue;
            double *pj = pset + (size_t)j * stride, *xj = xset + (size_t)j * stride;
            T p = beta_j[j] * (SNT ? ldnt<T>(pj, i) : ld<T>(pj, i));   // (:806)
            p = p + cp[j] * ro;                                  // (:807)
            T x = (SNT ? ldnt<T>(xj, i) : ld<T>(xj, i)) + cx[j] * q;   // (:834)
            x = x + alpha_j[j] * p;                              // (:835)
            if (SNT) stnt(xj, i, x); else st(xj, i, x);
            p = p + c1[j] * q;                                   // (:836)
            p = p + c2[j] * ro;                                  // (:837)
            if (SNT) stnt(pj, i, p); else st(pj, i, p);
        }
        const T rr = q + (-omega) * y;                           // (:840)
        st(r, i, rr);
        const T tt = ld<T>(t, i) + (-alpha) * ld<T>(v, i);       // (:842)
        const T ww = y + (-omega) * tt;                          // (:843)
        st(w, i, ww);
        const T h = ld<T>(rh, i);
        acc[0] += hsum(rr * rr);
        acc[1] += hsum(h * rr);
        acc[2] += hsum(h * ww);
        acc[3] += hsum(h * ld<T>(s, i));
        acc[4] += hsum(h * ld<T>(z, i));
    }
};
void launch_shift_pipe2(const Vecs &v, double *p_set, double *x_set, uint32_t set_stride, int seed, const ShiftDev *H,
                        Scal *S, Reduce red, hipStream_t s)
{
    auto go = [&](auto f) {
        f.xs = x_set + (size_t)seed * set_stride; f.ps = p_set + (size_t)seed * set_stride;
        f.r = v.r; f.w = v.w; f.pset = p_set; f.xset = x_set; f.t = v.t; f.v = v.v; f.rh = v.rh; f.s = v.s; f.z = v.z;
        f.rold = v.ax; f.H = H; f.stride = set_stride;
        run_vec(f, v.n, S, red, s);
    };
    if (stream_sets()) go(FShPipe2<true>{}); else go(FShPipe2<false>{});
}

struct FShiftPSeed {  // p[seed] = beta p[seed] ; += r ; += (-beta omega) s                (:317-319)
    static constexpr int ND = 0;
    double *p; const double *r, *s; double beta, c;
    __device__ void load(const Scal *S) { beta = S->beta; c = -S->beta * S->omega; }
    template <class T> __device__ void apply(uint32_t i, double *) const
    {
        T pp = beta * ld<T>(p, i);
        pp = pp + 1.0 * ld<T>(r, i);
        pp = pp + c * ld<T>(s, i);
        st(p, i, pp);
    }
};
void launch_shift_pseed(const Vecs &v, double *p_seed, Scal *S, hipStream_t s)
{
    run_vec(FShiftPSeed{p_seed, v.r, v.s, 0.0, 0.0}, v.n, S, Reduce{}, s);
}

struct FDrift {     // how far the recursive residual has drifted from the true one (adaptive replacement)
    static constexpr int ND = 2;
    static constexpr int kModes = kAnyMode;
    const double *b, *ax, *r;
    __device__ void load(const Scal *) {}
    template <class T> __device__ void apply(uint32_t i, double *acc) const
    {
        const T rr = ld<T>(r, i);
        const T dlt = (ld<T>(b, i) + (-1.0) * ld<T>(ax, i)) - rr;
        acc[0] += hsum(dlt * dlt);
        acc[1] += hsum(rr * rr);
    }
};
void launch_drift(const Vecs &v, const Launch &L, Reduce red) { run_vec(FDrift{v.b, v.ax, v.r}, v.n, L, red); }

struct FDot {
    static constexpr int ND = 1;
    const double *x, *y;
    __device__ void load(const Scal *) {}
    template <class T> __device__ void apply(uint32_t i, double *acc) const { acc[0] += hsum(ld<T>(x, i) * ld<T>(y, i)); }
};
void launch_dot(const double *x, const double *y, uint32_t n, Scal *S, Reduce red, hipStream_t s)
{
    run_vec(FDot{x, y}, n, S, red, s);
}

#endif
}  // namespace bicg
