// bicg_jagw.hip -- the product of a block with RAGGED rows (jagged slices + x window in LDS, SellDev::win_*: the layout an
// unstructured FEM matrix such as Transport.mtx gets) for one rank without halo, with the dependent memory trips of a 256-row
// group cut from twelve to three.
//
// What the counters of k_spmv_sell<.., LAY_JAGW, ..> said on the FEM-like matrix (profiles/r05/fem_like_spmv_counters_before.txt):
// waves waiting 64 % of their cycles, 658 cycles per L1 miss, the units busy 10 % -- and the work of a group is a CHAIN: group
// number, window bounds, then per run of the window {descriptor, x values, LDS stores} one after the other, barrier, slice
// metadata, row pointers of the lane's row (behind the row permutation), first batch of entries, second batch, ... Every link is
// a round trip of 0.5-1 us; a workgroup lives for a dozen of them and the product's time is rounds of workgroups times that chain,
// not bytes over bandwidth (4.5-4.8 TB/s of its 276 MB). Here:
//   trip 1   window bounds, slice base / length (scalar), ONE 16-bit word per lane = its row in the group and that row's length
//            (SellDev::lane_info: no row pointers, no separate permutation load);
//   trip 2   every run descriptor of the window at once (one vector load, lane r = run r), the first TWO batches of the lane's
//            entries (values + 16-bit window slots), the dot operand;
//   trip 3   all x values of the window (<= 8 per thread, the slot -> column search runs over the run descriptors in registers),
//            then the LDS stores, ONE barrier, and the gathers; further batches are requested two ahead of their use.
// Arithmetic: each lane adds ITS row's products in stored order, one rounding per product and per sum, y_i = 0.0 + that sum --
// bit for bit the sum of mult() (reference src/matrix.c:506-515) and of k_spmv_sell. The fused dots and their reduction are those
// of k_spmv_sell (same slots, same order: the solvers' scalars do not change in a single bit against that kernel).
#include "bicg_device.h"
#include "bicg_devfn.h"
#include "bicg_reduce.h"

#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

namespace bicg {

#define BICG_KCONST __attribute__((address_space(4)))
extern __shared__ double jagw_win[];      // the x window of the group (SellDev::win_slots doubles)

constexpr int kJagU = 8;                  // entries per lane and batch
constexpr int kJagSlots = 8;              // window values per thread the kernel stages (win_slots <= 256 * kJagSlots)
constexpr uint32_t kJagMaxRuns = kJagwMaxRuns;      // run descriptors one wavefront can hold (lane r = run r)
static_assert(kJagwMaxSlots == (uint32_t)(kBlock * kJagSlots), "window values staged per group");

struct JagBatch { double v[kJagU]; uint32_t s[kJagU]; };

// the lane's entries k0 .. k0 + 7 of its row: step k of a jagged slice stores the entries of the lanes whose row is longer
// than k, in lane order (ballot + population count, all from the row lengths: every load of the batch is issued back to back)
template <bool NT>
__device__ __forceinline__ void jag_load(const SpmvArgs &a, uint32_t k0, uint32_t mylen, uint32_t &pos, JagBatch &B)
{
    const unsigned short *sl = reinterpret_cast<const unsigned short *>(a.sell.col16);
#pragma unroll
    for (int e = 0; e < kJagU; ++e) {
        const bool mine = k0 + (uint32_t)e < mylen;
        const unsigned long long m = __ballot(mine);
        const uint32_t j = pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        pos += (uint32_t)__builtin_popcountll(m);
        B.s[e] = 0u; B.v[e] = 0.0;
        if (mine) {       // (nothing that depends on a loaded value inside the predicated block: see k_spmv_sell)
            B.s[e] = NT ? __builtin_nontemporal_load(sl + j) : sl[j];
            B.v[e] = NT ? __builtin_nontemporal_load(a.sell.val + j) : a.sell.val[j];
        }
    }
}
__device__ __forceinline__ double jag_use(const JagBatch &B, uint32_t k0, uint32_t mylen, const double *win, double sum)
{
    double xv[kJagU];
#pragma unroll
    for (int e = 0; e < kJagU; ++e) xv[e] = win[B.s[e]];          // slot 0 for a lane whose row has ended
#pragma unroll
    for (int e = 0; e < kJagU; ++e)
        if (k0 + (uint32_t)e < mylen) sum += B.v[e] * xv[e];      // stored order
    return sum;
}

// Registers and occupancy (profiles/r05/fem_like_waves_per_simd.txt). The kernel used 97 registers without dots and 102-104 with
// them: just above the step at 96 where a SIMD holds five wavefronts instead of four. Forced below it the variants with dots
// spilled five words per lane, and a kernel that needs scratch at all starts its wavefronts slower (plain 0.141 -> 0.166 ms).
// Seven of those registers held tid + 256 k, hoisted out of the group loop; re-formed per group (an opaque copy of tid) every
// variant fits into 89-95 registers without scratch: product back to back 45.1 -> 42.3 us, plain iteration 0.141 -> 0.134 ms,
// CA 0.154 -> 0.151, pipelined 0.160 -> 0.155 on the FEM-like matrix.
template <int NDOT, bool NT, int MODE, bool LIST>
__device__ __forceinline__ void jagw_body(const SpmvArgs &a)
{
    constexpr int ND = NDOT > 0 ? NDOT : 1;
    const int done = a.S->done;
    __shared__ double sm[5 * ND];
    const unsigned bid = blockIdx.x, nblocks = gridDim.x;
    if (MODE == RED_WAVE) {
        __shared__ FinishLds fl;
        if (a.fin.seq && (bid < (unsigned)kShards || (a.fin.roles & FIN_APPLY))) (void)finish_group(a.S, a.fin, a.fin.roles, bid, nblocks, fl, nullptr);
    }
    double acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = 0.0;

    // groups of this workgroup: the order, placement and slot rules of k_spmv_sell
    unsigned vb = bid;
    if (a.xcd_map && bid < (nblocks / 8u) * 8u) vb = (bid % 8u) * (nblocks / 8u) + bid / 8u;
    if (a.reverse) vb = nblocks - 1u - vb;
    const unsigned each = (a.nlist + nblocks - 1u) / nblocks;
    const unsigned gfirst = vb * each, gend = gfirst + each < a.nlist ? gfirst + each : a.nlist;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const double *__restrict__ x = a.x;
    double *const win = jagw_win;

    for (unsigned gq = gfirst; gq < gend && !done; ++gq) {
        const unsigned gi = a.reverse ? gfirst + (gend - 1u - gq) : gq;
        const unsigned g = a.glist ? a.glist[gi] : gi;
        // ---- trip 1 (LIST: the group's slots are positions in the list of its distinct columns, SellDev::win_list)
        const uint32_t w0 = LIST ? a.sell.win_lptr[g] : a.sell.win_ptr[g], w1 = LIST ? a.sell.win_ltotal[g] : a.sell.win_ptr[g + 1];
        // (the block's last group may lack its last slices: their metadata does not exist -- no rows, no entries)
        const uint32_t slice = g * (kGroupRows / kSliceRows) + wave;
        const bool has_slice = slice * (uint32_t)kSliceRows < a.nrows;
        const uint32_t base = has_slice ? a.sell.slice_base[slice] : 0u, len = has_slice ? a.sell.slice_len[slice] : 0u;
        const uint32_t info = a.sell.lane_info[(size_t)g * kGroupRows + tid];
        const uint32_t row = g * kGroupRows + (info & 0xFFu), mylen = info >> 8;          // (rows past the block's last: length 0)
        const bool live = row < a.nrows;
        // ---- trip 2
        const uint32_t nruns = LIST ? 0u : w1 - w0;
        uint2 myrun = make_uint2(0u, 0u);
        if (!LIST && lane < nruns) myrun = a.sell.win_runs[w0 + lane];
        uint32_t lcol[LIST ? kJagSlots / 2 : 1];                   // LIST: the columns of this thread's slots, two 16-bit distances per word
        if (LIST) {
#pragma unroll
            for (int j = 0; j < kJagSlots / 2; ++j) {
                lcol[j] = 0u;
                if (tid + (uint32_t)(2 * j) * kBlock < w1) lcol[j] = a.sell.win_list[w0 + (uint32_t)j * kBlock + tid];
            }
        }
        uint32_t pos = base;
        JagBatch A, B;
        jag_load<NT>(a, 0u, mylen, pos, A);
        jag_load<NT>(a, (uint32_t)kJagU, mylen, pos, B);
        double upre = 0.0, xown = 0.0;
        if (NDOT >= 1 && live) upre = a.u[row];
        if (a.has_shift && live) xown = x[row];
        // ---- trip 3: the window. Thread t stages slots t, t + 256, ...; slot -> column over the run descriptors
        const uint32_t last = nruns ? nruns - 1u : 0u;
        const uint32_t lrun_y = (uint32_t)__builtin_amdgcn_readlane((int)myrun.y, (int)last);
        const uint32_t total = LIST ? w1 : (nruns ? (lrun_y >> 16) + (lrun_y & 0xFFFFu) : 0u);
        double xw[kJagSlots];
        uint32_t col[kJagSlots];
        uint32_t ts = tid;
        asm volatile("" : "+v"(ts));       // (the slots tid + 256 k are re-formed per group: hoisted out of the loop they hold seven registers)
#pragma unroll
        for (int k = 0; k < kJagSlots; ++k)
            col[k] = LIST ? g * kGroupRows + (uint32_t)(int)(short)((k & 1) ? lcol[LIST ? k / 2 : 0] >> 16 : lcol[LIST ? k / 2 : 0] & 0xFFFFu)
                          : (uint32_t)__builtin_amdgcn_readlane((int)myrun.x, 0) + ts + (uint32_t)k * kBlock;
        for (uint32_t r = 1; r < nruns; ++r) {                    // wave-uniform: runs are in ascending slot order
            const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)myrun.x, (int)r);
            const uint32_t slot0 = (uint32_t)__builtin_amdgcn_readlane((int)myrun.y, (int)r) >> 16;
#pragma unroll
            for (int k = 0; k < kJagSlots; ++k) {
                const uint32_t s = ts + (uint32_t)k * kBlock;
                if (s >= slot0) col[k] = first + (s - slot0);
            }
        }
#pragma unroll
        for (int k = 0; k < kJagSlots; ++k) {
            const uint32_t s = ts + (uint32_t)k * kBlock;
            xw[k] = x[s < total ? col[k] : row < a.nrows ? row : 0u];       // (unconditional: a slot past the window reads a value that exists)
        }
        __syncthreads();                                          // the previous group's reads of the window are done
#pragma unroll
        for (int k = 0; k < kJagSlots; ++k) {
            const uint32_t s = ts + (uint32_t)k * kBlock;
            if (s < total) win[s] = xw[k];
        }
        __syncthreads();
        // ---- the rows: batches requested two ahead of their use, no value moved between registers
        double sum = 0.0;
        for (uint32_t k0 = 0; k0 < len; k0 += 2u * (uint32_t)kJagU) {
            sum = jag_use(A, k0, mylen, win, sum);
            if (k0 + (uint32_t)kJagU >= len) break;
            jag_load<NT>(a, k0 + 2u * (uint32_t)kJagU, mylen, pos, A);
            sum = jag_use(B, k0 + (uint32_t)kJagU, mylen, win, sum);
            if (k0 + 2u * (uint32_t)kJagU >= len) break;
            jag_load<NT>(a, k0 + 3u * (uint32_t)kJagU, mylen, pos, B);
        }
        double yi = 0.0 + sum;                                    // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
        if (a.has_shift && live) yi += a.shift * xown;            // (A + sigma I) x, src/shifted_solver.c:260
        if (live) a.y[row] = yi;
        if (NDOT >= 1 && live) {
            acc[0] += upre * yi;
            if (NDOT == 2) acc[NDOT >= 2 ? 1 : 0] += yi * yi;
            if (NDOT == 3) acc[NDOT >= 2 ? 1 : 0] += upre * upre;
        }
    }
    if (NDOT > 0 && !done) {
        if (MODE == RED_WAVE) wave_publish<ND>(acc, a.red.partial, a.red.slot_base + vb);
        else reduce_publish<ND, MODE == RED_TICKET_HEAVY>(acc, a.S, a.red, a.red.slot_base + vb, sm, a.red.slot_base + bid);
    }
}

template <int NDOT, bool NT, int MODE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) k_spmv_jagw(SpmvArgs a) { jagw_body<NDOT, NT, MODE, false>(a); }
#ifndef JAGL_WAVES
#define JAGL_WAVES 5
#endif
// the list-driven window holds its slots' columns (four registers: two 16-bit distances each) while the first batches are in flight
template <int NDOT, bool NT, int MODE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(JAGL_WAVES, 8))) k_spmv_jagl(SpmvArgs a) { jagw_body<NDOT, NT, MODE, true>(a); }

// ---- the same product WITHOUT a window: x gathered through the caches ----------------------------------------------------------
// For numberings whose 256-row groups touch many short runs of columns (reverse Cuthill-McKee of a tetrahedral mesh: up to 170 runs
// per group; a random permutation: one column per run) a window has nothing to stage in bulk, and the plan keeps the jagged slices
// with 16-bit offsets from the row (or 32-bit columns when some entry is further than 32 767 from its row). k_spmv_sell walks
// such a slice batch by batch -- row pointers, then per batch {entries, then gathers}: seven dependent trips for a 20-entry row.
// Here, as in k_spmv_jagw:
//   trip 1   slice base / length (scalar), one 16-bit word per lane (its row's length);
//   trip 2   the first TWO batches (of six) of the lane's entries (values + offsets / columns), the offsets / columns of the THIRD batch,
//            the dot operand;
//   trip 3   the 12 gathers of the first two batches;
//   trip 4   (slices longer than 12) the third batch's values and gathers together -- its columns arrived with trip 2;
// longer slices (> 18) continue batch by batch.
// Same sums in the same order as mult() (reference src/matrix.c:506-515) and as k_spmv_sell; dots and their reduction likewise.
// Batches of kJagdU = 6 entries: with 8 the kernel needs more than the 96 registers a wavefront may hold at five per SIMD and
// spills (107 us per product on the RCM-numbered mesh matrix); 8 entries at four wavefronts 56.0 us, 8 without the third batch's
// column prefetch 55.8, 4 entries 56.7, 6 entries 53.5 us (k_spmv_sell's loop: 56.6) -- profiles/r06/mesh_probe_jagd_variants.txt,
// tools/jagd_variants.sh. The three-trip form gains far less here than with the window (47.9 -> 41.7 us there): what bounds this
// product is the gather path itself -- 64 lanes of a step touch dozens of cache lines -- not the length of the dependency chain.
// Addresses: every load of this kernel is a BUFFER load -- a scalar resource (base of the slice's entries / of x) and ONE 32-bit
// byte offset per lane instead of a 64-bit address pair: with flat addresses the 40 loads in flight took 30-50 registers more than
// the 96 a wavefront may hold at five per SIMD, and spilled.
#ifndef JAGD_U
#define JAGD_U 6
#endif
#ifndef JAGD_WAVES
#define JAGD_WAVES 5
#endif
#ifndef JAGD_PREFETCH
#define JAGD_PREFETCH 1
#endif
constexpr int kJagdU = JAGD_U;            // entries per lane and batch of k_spmv_jagd
struct JagCols { uint32_t c[kJagdU]; };
struct JagVals { double v[kJagdU]; };
typedef unsigned int jag_u32x2 __attribute__((ext_vector_type(2)));
struct JagRes { __amdgpu_buffer_rsrc_t val, col; };
__device__ __forceinline__ __amdgpu_buffer_rsrc_t jag_rsrc(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0xFFFFFFFFu, 0x00020000);      // raw buffer, no bounds in the way
}

// entries k0 .. k0 + kJagdU - 1 of the lane's row; pos = entries of the slice before step k0 (wave-uniform)
template <bool NT, bool C16, bool VALS, bool COLS>
__device__ __forceinline__ void jagd_load(const JagRes &R, uint32_t k0, uint32_t mylen, uint32_t &pos, JagVals &V, JagCols &C)
{
#pragma unroll
    for (int e = 0; e < kJagdU; ++e) {
        const bool mine = k0 + (uint32_t)e < mylen;
        const unsigned long long m = __ballot(mine);
        const uint32_t j = pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        pos += (uint32_t)__builtin_popcountll(m);
        if (COLS) C.c[e] = 0u;
        if (VALS) V.v[e] = 0.0;
        if (mine) {       // (nothing that depends on a loaded value inside the predicated block)
            if (COLS) {
                if (C16) C.c[e] = (uint32_t)(int)(short)__builtin_amdgcn_raw_buffer_load_b16(R.col, j * 2u, 0, NT ? 2 : 0);
                else C.c[e] = __builtin_amdgcn_raw_buffer_load_b32(R.col, j * 4u, 0, NT ? 2 : 0);
            }
            if (VALS) V.v[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(R.val, j * 8u, 0, NT ? 2 : 0));
        }
    }
}
// x of the batch's columns; a lane whose row has ended reads x of its own row (unpredicated: see k_spmv_sell)
template <bool C16>
__device__ __forceinline__ void jagd_gather(__amdgpu_buffer_rsrc_t xr, const JagCols &C, uint32_t k0, uint32_t mylen, uint32_t rb, JagVals &X)
{
#pragma unroll
    for (int e = 0; e < kJagdU; ++e) {
        const uint32_t col = C16 ? rb + C.c[e] : (k0 + (uint32_t)e < mylen ? C.c[e] : rb);
        X.v[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xr, col * 8u, 0, 0));
    }
}
__device__ __forceinline__ double jagd_use(const JagVals &V, const JagVals &X, uint32_t k0, uint32_t mylen, double sum)
{
#pragma unroll
    for (int e = 0; e < kJagdU; ++e)
        if (k0 + (uint32_t)e < mylen) sum += V.v[e] * X.v[e];      // stored order
    return sum;
}

template <int NDOT, bool NT, int MODE, bool C16>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(JAGD_WAVES, 8))) k_spmv_jagd(SpmvArgs a)
{
    constexpr int ND = NDOT > 0 ? NDOT : 1;
    constexpr uint32_t U = (uint32_t)kJagdU;
    const int done = a.S->done;
    __shared__ double sm[5 * ND];
    const unsigned bid = blockIdx.x, nblocks = gridDim.x;
    if (MODE == RED_WAVE) {
        __shared__ FinishLds fl;
        if (a.fin.seq && (bid < (unsigned)kShards || (a.fin.roles & FIN_APPLY))) (void)finish_group(a.S, a.fin, a.fin.roles, bid, nblocks, fl, nullptr);
    }
    double acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = 0.0;

    // groups of this workgroup: the order, placement and slot rules of k_spmv_sell
    unsigned vb = bid;
    if (a.xcd_map && bid < (nblocks / 8u) * 8u) vb = (bid % 8u) * (nblocks / 8u) + bid / 8u;
    if (a.reverse) vb = nblocks - 1u - vb;
    const unsigned each = (a.nlist + nblocks - 1u) / nblocks;
    const unsigned gfirst = vb * each, gend = gfirst + each < a.nlist ? gfirst + each : a.nlist;
    const unsigned tid = threadIdx.x, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const double *__restrict__ x = a.x;
    const __amdgpu_buffer_rsrc_t xr = jag_rsrc(a.x);

    for (unsigned gq = gfirst; gq < gend && !done; ++gq) {
        const unsigned gi = a.reverse ? gfirst + (gend - 1u - gq) : gq;
        const unsigned g = a.glist ? a.glist[gi] : gi;
        // ---- trip 1
        const uint32_t slice = g * (kGroupRows / kSliceRows) + wave;
        const bool has_slice = slice * (uint32_t)kSliceRows < a.nrows;
        const uint32_t base = has_slice ? a.sell.slice_base[slice] : 0u, len = has_slice ? a.sell.slice_len[slice] : 0u;
        const uint32_t info = a.sell.lane_info[(size_t)g * kGroupRows + tid];
        const uint32_t row = g * kGroupRows + (info & 0xFFu), mylen = info >> 8;          // (rows past the block's last: length 0)
        const bool live = row < a.nrows;
        const uint32_t rb = live ? row : 0u;
        // ---- trip 2
        JagRes R;
        R.val = jag_rsrc(a.sell.val + base);
        R.col = C16 ? jag_rsrc(a.sell.col16 + base) : jag_rsrc(a.sell.col + base);
        uint32_t pos = 0u;                                       // entries of the slice before the current step
        JagVals V0, V1, X0, X1;
        JagCols C0, C1, C2, C3;
        jagd_load<NT, C16, true, true>(R, 0u, mylen, pos, V0, C0);
        jagd_load<NT, C16, true, true>(R, U, mylen, pos, V1, C1);
        const uint32_t pos2 = pos;
        if (JAGD_PREFETCH >= 1 && len > 2u * U) jagd_load<NT, C16, false, true>(R, 2u * U, mylen, pos, V0, C2);
        const uint32_t pos3 = pos;
        if (JAGD_PREFETCH >= 2 && len > 3u * U) jagd_load<NT, C16, false, true>(R, 3u * U, mylen, pos, V0, C3);
        double upre = 0.0, xown = 0.0;
        if (NDOT >= 1 && live) upre = a.u[row];
        if (a.has_shift && live) xown = x[row];
        // ---- trip 3
        jagd_gather<C16>(xr, C0, 0u, mylen, rb, X0);
        jagd_gather<C16>(xr, C1, U, mylen, rb, X1);
        double sum = jagd_use(V0, X0, 0u, mylen, 0.0);
        if (JAGD_PREFETCH >= 1 && len > 2u * U) {
            // ---- trip 4: values and gathers of the third batch in one trip (its columns arrived with trip 2)
            uint32_t p = pos2;
            jagd_load<NT, C16, true, false>(R, 2u * U, mylen, p, V0, C0);
            jagd_gather<C16>(xr, C2, 2u * U, mylen, rb, X0);
        }
        sum = jagd_use(V1, X1, U, mylen, sum);
        if (JAGD_PREFETCH >= 2 && len > 3u * U) {                    // ... and of the fourth, behind it in the same trip
            uint32_t p = pos3;
            jagd_load<NT, C16, true, false>(R, 3u * U, mylen, p, V1, C1);
            jagd_gather<C16>(xr, C3, 3u * U, mylen, rb, X1);
        }
        if (JAGD_PREFETCH >= 1 && len > 2u * U) sum = jagd_use(V0, X0, 2u * U, mylen, sum);
        if (JAGD_PREFETCH >= 2 && len > 3u * U) sum = jagd_use(V1, X1, 3u * U, mylen, sum);
        for (uint32_t k0 = (2u + (uint32_t)JAGD_PREFETCH) * U; k0 < len; k0 += U) {      // longer slices: batch by batch
            jagd_load<NT, C16, true, true>(R, k0, mylen, pos, V1, C1);
            jagd_gather<C16>(xr, C1, k0, mylen, rb, X1);
            sum = jagd_use(V1, X1, k0, mylen, sum);
        }
        double yi = 0.0 + sum;                                    // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
        if (a.has_shift && live) yi += a.shift * xown;            // (A + sigma I) x, src/shifted_solver.c:260
        if (live) a.y[row] = yi;
        if (NDOT >= 1 && live) {
            acc[0] += upre * yi;
            if (NDOT == 2) acc[NDOT >= 2 ? 1 : 0] += yi * yi;
            if (NDOT == 3) acc[NDOT >= 2 ? 1 : 0] += upre * upre;
        }
    }
    if (NDOT > 0 && !done) {
        if (MODE == RED_WAVE) wave_publish<ND>(acc, a.red.partial, a.red.slot_base + vb);
        else reduce_publish<ND, MODE == RED_TICKET_HEAVY>(acc, a.S, a.red, a.red.slot_base + vb, sm, a.red.slot_base + bid);
    }
}

// jagged slices without a window, one rank's halo-free rows, the per-lane words present
bool jagd_fast_ok(const SpmvArgs &a, bool with_offd, bool fused_halo)
{
    // (x is addressed by 32-bit byte offsets: fewer than 2^28 rows)
    return a.sell.jag && a.sell.win_slots == 0 && a.sell.lane_info != nullptr && !with_offd && !fused_halo &&
           (a.sell.col16 != nullptr || a.sell.col != nullptr) && a.nrows < (1u << 28);
}

template <class K>
static void jagd_go(K kernel, const SpmvArgs &a, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    const dim3 g(sell_grid(a.nlist, a.groups_per_wg)), b(kBlock);
    if (e0 && e1) hipExtLaunchKernelGGL(kernel, g, b, 0, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kernel, g, b, 0, st, a);
    static const bool debug = getenv("BICG_DEBUG") != nullptr;
    if (debug) {
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) fprintf(stderr, "bicgstab_hip: HIP error \"%s\" noticed at: k_spmv_jagd\n", hipGetErrorString(err));
    }
}

bool launch_spmv_jagd(const SpmvArgs &a, int ndot, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    if (a.nlist == 0) return false;
    g_product_kernels |= PK_JAGD;
    const bool nt = a.nt != 0, c16 = a.sell.col16 != nullptr;
    const int mode = red_mode(a.red, a.fin, ndot > 0);
#define JAGD_C16(ND, MD, NTV)                                                             \
    do {                                                                                  \
        if (c16) jagd_go(k_spmv_jagd<ND, NTV, MD, true>, a, st, e0, e1);                  \
        else jagd_go(k_spmv_jagd<ND, NTV, MD, false>, a, st, e0, e1);                     \
    } while (0)
#define JAGD_MODE(ND, MD)                                                                 \
    do {                                                                                  \
        if (nt) JAGD_C16(ND, MD, true); else JAGD_C16(ND, MD, false);                     \
    } while (0)
#define JAGD_CASE(ND)                                                                     \
    do {                                                                                  \
        if (mode == RED_WAVE) JAGD_MODE(ND, RED_WAVE);                                    \
        else if (mode == RED_TICKET_HEAVY) JAGD_MODE(ND, ((ND) > 0 ? RED_TICKET_HEAVY : RED_TICKET)); \
        else JAGD_MODE(ND, RED_TICKET);                                                   \
    } while (0)
    if (ndot == 0) JAGD_CASE(0); else if (ndot == 1) JAGD_CASE(1); else if (ndot == 2) JAGD_CASE(2); else JAGD_CASE(3);
#undef JAGD_CASE
#undef JAGD_MODE
#undef JAGD_C16
    return true;
}

// can this launch go to k_spmv_jagw? (one rank's halo-free rows, the plan's per-lane words present, the window small enough)
bool jagw_fast_ok(const SpmvArgs &a, bool with_offd, bool fused_halo)
{
    return a.sell.win_slots > 0 && a.sell.lane_info != nullptr && !with_offd && !fused_halo &&
           a.sell.win_slots <= (uint32_t)(kBlock * kJagSlots) && a.sell.win_max_runs >= 1 &&
           (a.sell.win_max_runs <= kJagMaxRuns || a.sell.win_list != nullptr);
}

template <class K>
static void jagw_go(K kernel, const SpmvArgs &a, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    const dim3 g(sell_grid(a.nlist, a.groups_per_wg)), b(kBlock);
    const unsigned lds = a.sell.win_slots * (unsigned)sizeof(double);
    if (e0 && e1) hipExtLaunchKernelGGL(kernel, g, b, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kernel, g, b, lds, st, a);
    static const bool debug = getenv("BICG_DEBUG") != nullptr;
    if (debug) {
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) fprintf(stderr, "bicgstab_hip: HIP error \"%s\" noticed at: k_spmv_jagw\n", hipGetErrorString(err));
    }
}

bool launch_spmv_jagw(const SpmvArgs &a, int ndot, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    if (a.nlist == 0) return false;
    const bool list = a.sell.win_list != nullptr;
    g_product_kernels |= list ? PK_JAGW_LIST : PK_JAGW;
    const bool nt = a.nt != 0;
    const int mode = red_mode(a.red, a.fin, ndot > 0);
#define JAGW_MODE(ND, MD)                                                                 \
    do {                                                                                  \
        if (list) { if (nt) jagw_go(k_spmv_jagl<ND, true, MD>, a, st, e0, e1); else jagw_go(k_spmv_jagl<ND, false, MD>, a, st, e0, e1); } \
        else if (nt) jagw_go(k_spmv_jagw<ND, true, MD>, a, st, e0, e1);                   \
        else jagw_go(k_spmv_jagw<ND, false, MD>, a, st, e0, e1);                          \
    } while (0)
#define JAGW_CASE(ND)                                                                     \
    do {                                                                                  \
        if (mode == RED_WAVE) JAGW_MODE(ND, RED_WAVE);                                    \
        else if (mode == RED_TICKET_HEAVY) JAGW_MODE(ND, ((ND) > 0 ? RED_TICKET_HEAVY : RED_TICKET)); \
        else JAGW_MODE(ND, RED_TICKET);                                                   \
    } while (0)
    if (ndot == 0) JAGW_CASE(0); else if (ndot == 1) JAGW_CASE(1); else if (ndot == 2) JAGW_CASE(2); else JAGW_CASE(3);
#undef JAGW_CASE
#undef JAGW_MODE
    return true;
}

void preload_jagw_kernels()
{
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k_spmv_jagw<0, false, RED_TICKET>));
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k_spmv_jagd<0, false, RED_TICKET, true>));
    (void)hipGetLastError();
}

}  // namespace bicg
