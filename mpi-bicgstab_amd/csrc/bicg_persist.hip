// bicg_persist.hip -- the pipelined BiCGStab iteration (reference src/solver.c:351-398) as ONE persistent launch for
// latency-bound ranks: a 200 k-row rank (1/8 of Transport) holds 30 MB of matrix -- 117 KB per CU, which fits the
// 160 KB of LDS of a CDNA4 CU -- and 10 work vectors of one value per thread. So a workgroup of up to 1024 threads owns
// the same <= 1024 rows for a whole chunk of iterations: matrix slices and the x window in LDS, the vectors in
// registers, and the only traffic per iteration is what other workgroups (or other GPUs) must see, sent as LL words.
// See struct PersistArgs (bicg_device.h) for the protocol. The multi-launch form of the same iteration
// (k_spmv_sell_epi, two launches of ~13 us on such a rank) pays a kernel boundary per SpMV; here an SpMV costs one
// neighbour hand-off (~1-2 us) plus LDS arithmetic.
//
// Arithmetic: every expression is the one of FPipe1 / FPipe2 / sell_row (bicg_kernels.hip), operation for operation,
// with -ffp-contract=off: rows and element-wise phases are bit-identical to the multi-launch path and to the
// reference; the dot sums are associated differently (wavefront -> workgroup -> table in workgroup order), fixed, so
// runs are bit-reproducible.
#include "bicg_device.h"
#include "bicg_devfn.h"

#include <cstdio>
#include <cstdlib>
#include <set>
#include <utility>

namespace bicg {

namespace {

constexpr int kMaxWaves = 16;             // 15 row wavefronts + the communication wavefront

struct PersistLds {
    Scal   priv;                          // helper: the scalar block the recurrence runs on
    double wsum[kMaxWaves * kMaxDots];    // per-wavefront partial sums
    double sums[kRedSlots];
    double sc[4];                         // alpha, beta, omega, done as received
    double pv[kRedSlots * kMaxRanksP2p];  // helper, multi rank: every rank's sums
    int    fail;
    int    drift_flag;                    // helper: the drift group just applied asks for a replacement iteration
    int    adaptive;                      // helper: replacement iterations asked for by the drift check in this launch
    double coef[6 * kPersistMaxShifts];   // shifted kernel: beta_j, alpha_j, cp, cx, c1, c2 of the iteration as received
};

// Workgroup barrier that orders LDS traffic only. __syncthreads() also waits for the wavefront's outstanding GLOBAL
// accesses, and nothing inside a workgroup is ever handed over through global memory here.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ bool alarm_raised(const int *alarm)
{
    return __hip_atomic_load(alarm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void raise_alarm(int *alarm)
{
    __hip_atomic_store(alarm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// An LL pair (one double) as ONE 16-byte access: {payload lo, tag, payload hi, tag}. Each 8-byte half carries its own
// tag, so a torn access is recognisable. (The s_nop after the store: a VALU write to the data registers of a store wider
// than 64 bits needs wait states the compiler inserts for its own stores but cannot know about here.) Write-through / L1-bypassing (sc0 sc1: system scope, also right for the
// uncached landing rings other GPUs store into). Scalar 8-byte write-through stores cost ~3 x the fabric time per byte.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ll_store16(llword *dst, double v, unsigned tag)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const u32x4 w = {(unsigned)bits, tag, (unsigned)(bits >> 32), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
}
__device__ __forceinline__ bool ll_decode(const u32x4 &w, unsigned tag, double *out)
{
    *out = __longlong_as_double((long long)(((unsigned long long)w.z << 32) | (unsigned long long)w.x));
    return w.y == tag && w.w == tag;
}
// loads of 1 / 2 / 4 / 5 pairs, all requested before the first result is waited for (the waitcnt is part of the asm
// statement: the compiler never sees a register that is still in flight). SYS: system scope (sc0 sc1) for words another
// GPU stores into; agent scope (sc1) inside this GPU.
#define LL_LD(SYSV) (SYSV ? "sc0 sc1" : "sc1")
template <bool SYS> __device__ __forceinline__ void ll_load16_x1(const llword *p0, u32x4 &r0)
{
    if (SYS) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0) : "v"(p0) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0) : "v"(p0) : "memory");
}
template <bool SYS> __device__ __forceinline__ void ll_load16_x2(const llword *p0, const llword *p1, u32x4 &r0, u32x4 &r1)
{
    if (SYS) asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                          : "=&v"(r0), "=&v"(r1) : "v"(p0), "v"(p1) : "memory");
    else asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                      : "=&v"(r0), "=&v"(r1) : "v"(p0), "v"(p1) : "memory");
}
template <bool SYS> __device__ __forceinline__ void ll_load16_x4(const llword *const (&p)[4], u32x4 (&r)[4])
{
    if (SYS) asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                          "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                          : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
    else asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                      "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
}
__device__ __forceinline__ void ll_load16_x5(const llword *p, u32x4 (&r)[5])       // five consecutive pairs, inside this GPU
{
    asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %5, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %5, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %5, off offset:64 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]) : "v"(p) : "memory");
}
// stores inside this GPU: agent scope
__device__ __forceinline__ void ll_store16_agent(llword *dst, double v, unsigned tag)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const u32x4 w = {(unsigned)bits, tag, (unsigned)(bits >> 32), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
}

// The x window (row wavefronts): the columns this workgroup's rows touch, as LL pairs of sequence `seq` (local rows) or
// `hseq` (halo positions, landing ring slot hseq % kHaloRing). Every thread requests its values first and then re-reads
// only those whose tags have not arrived. Row wavefronts never store to global memory inside the iteration loop: on
// gfx9 loads and stores share one in-order counter, so a load issued behind a write-through store would not be seen
// before that store has been acknowledged (~2 us) -- all stores are the communication wavefront's.
template <bool MULTI>
__device__ __forceinline__ void stage_window(const PersistArgs &a, const uint2 *runs, unsigned nruns, unsigned nslots, const llword *src,
                                             unsigned seq, unsigned hseq, double *win, unsigned nrt, PersistLds &L, const double *zs,
                                             uint32_t row0, uint32_t nmine)
{
    constexpr int W = 4;                  // values per thread and round
    const unsigned tid = threadIdx.x;
    const llword *ring = a.ring + (size_t)(hseq % kHaloRing) * a.halo * 2;
    bool ok = true;
    for (unsigned base = 0; base < nslots && ok; base += W * nrt) {
        const llword *p[W];
        unsigned want[W];
        unsigned pend = 0u;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const unsigned s = base + tid + j * nrt;
            p[j] = src; want[j] = seq;
            if (s < nslots) {
                unsigned r = 0;
                while (r + 1 < nruns && (runs[r + 1].y >> 16) <= s) ++r;       // runs are few and ordered by slot
                const uint32_t col = runs[r].x + (s - (runs[r].y >> 16));
                if (col - row0 < nmine) { win[s] = zs[col - row0]; continue; }      // this workgroup's own rows: straight from LDS
                if (col < a.nrows) { p[j] = src + 2 * (size_t)col; }
                else { p[j] = ring + 2 * (size_t)(col - a.nrows); want[j] = hseq; }
                pend |= 1u << j;
            }
        }
        const unsigned long long t0 = wall_clock64();
        // Nothing can have arrived before the neighbours' stores have crossed the fabric (~1 us): polling from the first
        // cycle only fills the fabric with 200 k threads' failed reads and delays the very stores they wait for.
        if (base == 0) for (unsigned i = 0; i < a.first_sleep; ++i) __builtin_amdgcn_s_sleep(8);
        for (unsigned spin = 0; pend; ++spin) {
            if (spin) __builtin_amdgcn_s_sleep(6);
            u32x4 w[W];
            ll_load16_x4<MULTI>(p, w);
#pragma unroll
            for (int j = 0; j < W; ++j) {
                double val;
                if (((pend >> j) & 1u) && ll_decode(w[j], want[j], &val)) {
                    win[base + tid + j * nrt] = val;
                    pend &= ~(1u << j);
                }
            }
            if (pend && (spin & 15u) == 15u) {
                if (wall_clock64() - t0 > a.timeout_ticks || alarm_raised(a.alarm)) { ok = false; break; }
            }
        }
    }
    if (!ok) { L.fail = 1; raise_alarm(a.alarm); }
}

// y_i of this lane's row: diag entries in stored order, then the offd entries (reference src/matrix.c:434-440, 506-515)
template <bool MULTI, int U = 8>
__device__ __forceinline__ double persist_row(const double *mval, const unsigned short *mslot, uint32_t slen, uint32_t mylen, uint32_t mydiag,
                                              const double *win)
{
    const unsigned lane = threadIdx.x & 63u;
    double sd = 0.0, so = 0.0;
    for (uint32_t k0 = 0; k0 < slen; k0 += U) {
        double v[U];
        unsigned sl[U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const bool in = k0 + e < slen;            // wave-uniform
            const uint32_t j = (k0 + e) * kSliceRows + lane;
            v[e] = in ? mval[j] : 0.0;
            sl[e] = in ? (unsigned)mslot[j] : 0u;
        }
        double xv[U];
#pragma unroll
        for (int e = 0; e < U; ++e) xv[e] = win[sl[e]];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            if (k0 + e < mydiag) sd += v[e] * xv[e];
            else if (MULTI && k0 + e < mylen) so += v[e] * xv[e];
        }
    }
    double yi = 0.0 + sd;
    if (MULTI) yi += so;
    return yi;
}

// row wavefronts: this row's value of the vector to publish and the wavefront's dot partials go to LDS
template <int N>
__device__ __forceinline__ void hand_over(double val, double (&acc)[N], double *zs, PersistLds &L, llword *img_row, unsigned seq)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // One 16-byte write-through store per row, all wavefronts at once (a single wavefront storing the whole image took
    // 1.6 us). The wavefront's next global loads -- the window -- cannot return before this store is acknowledged
    // (one in-order counter on gfx9), but nothing it waits for can be there earlier either.
    if (img_row) ll_store16_agent(img_row, val, seq);
    if (zs) zs[threadIdx.x] = val;
#pragma unroll
    for (int d = 0; d < N; ++d) acc[d] = wave_sum(acc[d]);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < N; ++d) L.wsum[wave * kMaxDots + d] = acc[d];
    }
}

// ... a value without dot partials
__device__ __forceinline__ void publish_only(double val, double *zs, llword *img_row, unsigned seq)
{
    if (img_row) ll_store16_agent(img_row, val, seq);
    zs[threadIdx.x] = val;
}

// communication wavefront: everything the rest of the GPU (and the other GPUs) gets from this workgroup in one phase --
// the LL image of its rows' values, the halo values other ranks need, its row of the dot table -- then the wait for the
// applied scalars of that group (needed by the row wavefronts only after their product)
template <int N>
__device__ __forceinline__ void comm_partials(unsigned lane, unsigned nrw, llword *tab_row, unsigned seq, PersistLds &L)
{
    if ((int)lane < N) {
        double t[kMaxWaves];
#pragma unroll
        for (int w = 0; w < kMaxWaves - 1; ++w) t[w] = (unsigned)w < nrw ? L.wsum[w * kMaxDots + lane] : 0.0;
        double s = t[0];
#pragma unroll
        for (int w = 1; w < kMaxWaves - 1; ++w) s += t[w];                          // wavefront order (+ 0.0 beyond the last)
        ll_store16_agent(tab_row + 2 * lane, s, seq);
    }
}
template <bool MULTI>
__device__ __forceinline__ void comm_halo(const PersistArgs &a, unsigned lane, const double *zs, unsigned hseq, unsigned ns0, unsigned ns1)
{
    if (MULTI)
        for (unsigned i = ns0 + lane; i < ns1; i += 64u)
            ll_store16(reinterpret_cast<llword *>(a.snd_dst0[i] + (unsigned long long)(hseq % kHaloRing) * a.snd_stride[i]),
                       zs[a.snd_row[i]], hseq);
}
template <int N, bool MULTI>
__device__ __forceinline__ void comm_phase(const PersistArgs &a, unsigned lane, uint32_t row0, uint32_t nmine, unsigned nrw, const double *zs,
                                           llword *img, unsigned seq, unsigned hseq, unsigned ns0, unsigned ns1, llword *tab_row,
                                           PersistLds &L)
{
    // the dot partials first: the helper's chain (table -> sums -> recurrence -> scalars back) is the longest of the phase
    comm_partials<N>(lane, nrw, tab_row, seq, L);
    comm_halo<MULTI>(a, lane, zs, hseq, ns0, ns1);
}

// ... and, once the row wavefronts are busy with their product, the wait for the applied scalars of that group
__device__ __forceinline__ void comm_scalars(const PersistArgs &a, unsigned lane, const llword *arow, unsigned seq, PersistLds &L)
{
    if (lane < 4) {
        double v = 0.0;
        const unsigned long long t0 = wall_clock64();
        for (unsigned spin = 0;; ++spin) {
            u32x4 w;
            ll_load16_x1<false>(arow + 2 * lane, w);
            if (ll_decode(w, seq, &v)) break;
            __builtin_amdgcn_s_sleep(4);
            if ((spin & 15u) == 15u && (wall_clock64() - t0 > a.timeout_ticks || alarm_raised(a.alarm))) {
                L.fail = 1; raise_alarm(a.alarm); v = 0.0;
                break;
            }
        }
        L.sc[lane] = v;
    }
}

// ... and (shifted kernel) the 6 x nsig coefficients of the iteration, published by the helper right after omega
__device__ __forceinline__ void comm_coefs(const PersistArgs &a, unsigned lane, const llword *crow, unsigned seq, PersistLds &L)
{
    for (int idx = (int)lane; idx < 6 * a.nsig; idx += 64) {
        const int q = idx / a.nsig, j = idx % a.nsig;
        const llword *src = crow + 2 * (size_t)(q * kPersistMaxShifts + j);
        double v = 0.0;
        const unsigned long long t0 = wall_clock64();
        for (unsigned spin = 0;; ++spin) {
            u32x4 w;
            ll_load16_x1<false>(src, w);
            if (ll_decode(w, seq, &v)) break;
            __builtin_amdgcn_s_sleep(2);
            if ((spin & 15u) == 15u && (wall_clock64() - t0 > a.timeout_ticks || alarm_raised(a.alarm))) {
                L.fail = 1; raise_alarm(a.alarm); v = 0.0;
                break;
            }
        }
        L.coef[q * kPersistMaxShifts + j] = v;
    }
}

// helper workgroup: sums of group `seq` over the table (workgroup order, fixed tree), exchanged with the other ranks,
// recurrence applied on L.priv, scalars published.
template <int N, bool SH = false>
__device__ __forceinline__ bool helper_group(const PersistArgs &a, const llword *tab, llword *row, unsigned seq, unsigned mseq, int phase,
                                             PersistLds &L, unsigned long long *stamp, double *smax = nullptr, llword *crow = nullptr)
{
#define HSTAMP(i) do { if (stamp && threadIdx.x == 0) stamp[i] = wall_clock64(); } while (0)
    const unsigned tid = threadIdx.x, nt = blockDim.x, lane = tid & 63u, wave = tid >> 6, nw = nt >> 6;
    double acc[N];
#pragma unroll
    for (int d = 0; d < N; ++d) acc[d] = 0.0;
    bool ok = true;
    for (unsigned g = tid; g < a.nwg && ok; g += nt) {
        const llword *src = tab + (size_t)g * kRedSlots * 2;
        double v[N];
        const unsigned long long t0 = wall_clock64();
        for (unsigned spin = 0;; ++spin) {
            bool all = true;
            if (N == 1) {
                u32x4 w0;
                ll_load16_x1<false>(src, w0);
                all = ll_decode(w0, seq, &v[0]);
            } else if (N == 2) {
                u32x4 w0, w1;
                ll_load16_x2<false>(src, src + 2, w0, w1);
                all = ll_decode(w0, seq, &v[0]) & ll_decode(w1, seq, &v[N > 1 ? 1 : 0]);
            } else {
                u32x4 w[5];
                ll_load16_x5(src, w);
#pragma unroll
                for (int d = 0; d < N; ++d) all = ll_decode(w[d], seq, &v[d]) && all;
            }
            if (all) break;
            __builtin_amdgcn_s_sleep(2);
            if ((spin & 15u) == 15u && (wall_clock64() - t0 > a.timeout_ticks || alarm_raised(a.alarm))) { ok = false; break; }
        }
        if (!ok) break;
#pragma unroll
        for (int d = 0; d < N; ++d) acc[d] += v[d];
    }
    if (!ok) { L.fail = 1; raise_alarm(a.alarm); }
    HSTAMP(0);
#pragma unroll
    for (int d = 0; d < N; ++d) acc[d] = wave_sum(acc[d]);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < N; ++d) L.wsum[wave * kMaxDots + d] = acc[d];
    }
    lds_barrier();
    HSTAMP(1);
    if (L.fail) return false;
    if ((int)tid < N) {
        double s = L.wsum[tid];
        for (unsigned w = 1; w < nw; ++w) s += L.wsum[w * kMaxDots + tid];
        L.sums[tid] = s;
    }
    lds_barrier();
    if (a.multi) {
        const unsigned long long t_mail = (!SH && a.waitlog && tid == 0) ? wall_clock64() : 0ull;
        // all-reduce over the ranks through the mailboxes (reference: one MPI_Iallreduce per dot, e.g. src/solver.c:363-367):
        // store this rank's sums into every rank's mailbox, wait for the P contributions, add them like a
        // recursive-doubling all-reduce would -- every rank gets the same bits. The stores are the LAST wavefront's, the
        // waits the first wavefronts': a wavefront's loads are not seen before its own earlier stores are acknowledged.
        const int P = a.p2p.nranks;
        if (wave == nw - 1)
            for (int t = lane; t < N * P; t += 64) {
                const int p = t / N, d = t % N;
                ll_store(a.p2p.mail[p] + mail_index(mseq, P, a.p2p.rank, d), L.sums[d], mseq);
            }
        const llword *mine = a.p2p.mail[a.p2p.rank];
        const unsigned npoll = nw > 1 ? (nw - 1) * 64u : 64u;
        if (tid < npoll)
            for (int t = tid; t < N * P; t += npoll) {
                const int p = t / N, d = t % N;
                double v;
                if (!ll_wait(mine + mail_index(mseq, P, p, d), mseq, a.timeout_ticks, &v)) L.fail = 1;
                L.pv[p * kRedSlots + d] = v;
            }
        lds_barrier();
        if (!SH && a.waitlog && tid == 0) a.waitlog[(seq - a.seq0 - 1u) % a.waitcap] = (unsigned)(wall_clock64() - t_mail) | 1u;
        if (L.fail) { if (tid == 0) raise_alarm(a.alarm); return false; }
        if ((int)tid < N) L.sums[tid] = rank_tree_sum(L.pv + tid, P);
        lds_barrier();
    }
    if (SH) {
        // shifted solvers: the seed's scalars by thread 0, the per-shift recurrences one thread per shift (bicg_devfn.h),
        // then every shift's coefficients of this iteration travel with omega as LL pairs (written by the thread that formed them)
        if (tid == 0) {
#pragma unroll
            for (int d = 0; d < N; ++d) L.priv.red[d] = L.sums[d];
        }
        __syncthreads();
        apply_phase_shifted<1024>(&L.priv, phase, smax);
        __syncthreads();
        if (crow) {
            const ShiftDev *H = L.priv.sh;
            for (int j = (int)tid; j < a.nsig; j += (int)nt) {
                const double cv[6] = {H->beta[j], H->alpha[j], H->cp[j], H->cx[j], H->c1[j], H->c2[j]};
#pragma unroll
                for (int q = 0; q < 6; ++q) ll_store16_agent(crow + 2 * (size_t)(q * kPersistMaxShifts + j), cv[q], seq);
            }
        }
    } else if (tid == 0) {
#pragma unroll
        for (int d = 0; d < N; ++d) L.priv.red[d] = L.sums[d];
        L.drift_flag = 0;
        if (phase == PH_DRIFT) {
            // adaptive residual replacement: ||(b - A x) - r|| > rr_drift ||r||  (the host's test of the multi-launch form,
            // Driver::drift, without the square root)
            if (L.sums[1] > 0.0 && L.sums[0] > a.drift_tol2 * L.sums[1]) { L.drift_flag = 1; L.adaptive += 1; }
        } else {
            apply_phase(&L.priv, phase, true);
        }
    }
    lds_barrier();
    HSTAMP(2);
    if (tid < 4) {
        // fourth value: 1 = the reference's loop has ended, 2 = the iteration that follows is a replacement iteration
        const double v = tid == 0 ? L.priv.alpha : tid == 1 ? L.priv.beta : tid == 2 ? L.priv.omega
                                  : L.priv.done ? 1.0 : L.drift_flag ? 2.0 : 0.0;
        ll_store16_agent(row + 2 * tid, v, seq);
    }
#undef HSTAMP
    return true;
}

// ---- what the three persistent kernels share ------------------------------------------------------------------
// Workgroup b runs on XCD b % 8 (observed placement, for speed only): give every XCD a contiguous range of rows, so that
// most of what a workgroup needs was published by a workgroup of its own XCD
__device__ __forceinline__ unsigned persist_wg(const PersistArgs &a)
{
    const unsigned b = blockIdx.x;
    return (a.xcd_map && b < (a.nwg / 8u) * 8u) ? (b % 8u) * (a.nwg / 8u) + b / 8u : b;
}

__device__ __forceinline__ unsigned blockIdxWg(const PersistArgs &a) { return persist_wg(a); }

// the helper workgroup's last act: the scalar block goes back to memory (the host reads k, (r,r), done from it)
__device__ __forceinline__ void helper_finish(const PersistArgs &a, PersistLds &L, double used_v = 0.0, double used_g = 0.0)
{
    lds_barrier();
    if (threadIdx.x == 0) {
        if (L.fail) { L.priv.done = 1; L.priv.comm_error = 1; }
        L.priv.red[kRedUsedV] = used_v; L.priv.red[kRedUsedG] = used_g; L.priv.red[kRedAdaptive] = (double)L.adaptive;
        *a.S = L.priv;
    }
}

// A row workgroup: spw row wavefronts (slices wg * spw ..., lane = row) + one communication wavefront. LDS: the window,
// the workgroup's matrix entries (LDSMAT), its window runs, the values it publishes in the current phase.
struct RowWg {
    double *win, *zs;
    const uint2 *runs;
    unsigned nrw, nrt, nruns, nslots, ns0, ns1;
    bool comm, live;
    uint32_t row0, nmine, row, slen, mylen, mydiag;
    const double *gval;
    const unsigned short *gslot;
    llword *tab0, *tab1, *img0, *img1;
};
template <bool LDSMAT, bool MULTI>
__device__ __forceinline__ RowWg row_setup(const PersistArgs &a, unsigned wg, double *dyn)
{
    const unsigned tid = threadIdx.x, nt = blockDim.x, lane = tid & 63u, wave = tid >> 6;
    RowWg R;
    R.win = dyn;
    double *mval = dyn + a.win_slots;
    unsigned short *mslot = reinterpret_cast<unsigned short *>(mval + a.mat_entries);
    uint2 *runs = reinterpret_cast<uint2 *>(mslot + ((a.mat_entries + 3u) & ~3u));
    R.runs = runs;
    R.zs = reinterpret_cast<double *>(runs + a.max_runs);          // [64 * spw] this workgroup's values of the vector being published
    R.nrw = a.spw; R.nrt = 64u * a.spw;
    R.comm = wave == R.nrw;
    const uint32_t s0 = wg * a.spw, s1 = min(a.nslices, s0 + a.spw);
    R.row0 = s0 * kSliceRows; R.nmine = min(a.nrows, s1 * kSliceRows) - R.row0;
    const uint32_t slice = s0 + wave;
    const bool have_slice = !R.comm && slice < a.nslices;
    R.row = slice * kSliceRows + lane;
    R.live = have_slice && R.row < a.nrows;
    const uint32_t e0 = a.pbase[s0], e1 = a.pbase[s1];
    uint32_t sbase = 0;
    R.slen = 0;
    if (have_slice) { sbase = a.pbase[slice]; R.slen = (a.pbase[slice + 1] - sbase) / kSliceRows; }
    R.mylen = R.live ? a.rlen[R.row] : 0u; R.mydiag = R.live ? a.rdiag[R.row] : 0u;
    const unsigned r0w = a.win_ptr[wg];
    R.nruns = a.win_ptr[wg + 1] - r0w;
    for (unsigned i = tid; i < R.nruns; i += nt) runs[i] = a.win_runs[r0w + i];
    R.nslots = 0;
    if (R.nruns) { const uint2 last = a.win_runs[r0w + R.nruns - 1]; R.nslots = (last.y >> 16) + (last.y & 0xFFFFu); }
    R.gval = a.pval + sbase; R.gslot = a.pslot + sbase;
    if (LDSMAT) {
        for (uint32_t j = e0 + tid; j < e1; j += nt) { mval[j - e0] = a.pval[j]; mslot[j - e0] = a.pslot[j]; }
        R.gval = mval + (sbase - e0); R.gslot = mslot + (sbase - e0);
    }
    R.ns0 = MULTI ? a.snd_ptr[wg] : 0u; R.ns1 = MULTI ? a.snd_ptr[wg + 1] : 0u;
    R.tab0 = a.dtab[0] + (size_t)wg * kRedSlots * 2; R.tab1 = a.dtab[1] + (size_t)wg * kRedSlots * 2;
    R.img0 = R.live ? a.llv[0] + 2 * (size_t)R.row : nullptr; R.img1 = R.live ? a.llv[1] + 2 * (size_t)R.row : nullptr;
    return R;
}

// ------------------------------------------------------------------------------------------------------------------
// Pipelined BiCGStab, R rows per thread.
//   R = 1: the form above -- 15 row wavefronts + the communication wavefront, 4 wavefronts per SIMD (128 registers each),
//          matrix slices in LDS when they fit (ranks up to ~245 k rows);
//   R = 2, 4, 8: ranks of 2 and 4 GPUs (400 k / 800 k rows of Transport = 1 565 / 3 130 rows per CU): 7 row wavefronts +
//          the communication wavefront = 2 wavefronts per SIMD (256 registers each), every thread keeps the ten vector
//          entries of its R rows in registers, the matrix slices are streamed from memory (they fit the Infinity Cache:
//          60 / 120 MB) and only the window and the published values live in LDS. The protocol between workgroups and
//          GPUs is the same; thread (wavefront w, lane l) owns rows (s0 + j nrw + w) 64 + l, j < R, i.e. positions
//          j nrt + tid of the workgroup's row range -- the index of its values in `zs`.
// Besides the plain iteration (src/solver.c:352-390) the kernel runs the residual-replacement iteration of
// pipe_bicgstab_rr (src/solver.c:494-548: p update, s = A p, z = A s, q / y and their dots, v = A z, x update, r = b - A x,
// w = A r, the five dots, t = A w -- six hand-offs instead of two, the same two dot groups) at the iterations the
// reference's schedule names (k % krr == 0, 0 < k <= krr nrr), at the first iteration of a launch when the host asks for
// it, and -- adaptive form, rr_drift > 0 -- whenever the true residual b - A x, recomputed every `drift_every` iterations
// with one more hand-off and a two-value group, has drifted from the recursive one by more than rr_drift (decided by the
// helper, so every workgroup and every rank takes the same branch).
// Sequence numbers run densely through a launch: hand-off n carries image tag vseq0 + n and halo exchange number
// halo_seq0 + n; dot group g carries table tag seq0 + g and mailbox number p2p.seq + g - 1 and uses table / scalar row
// g & 1. The helper reports how many of each the launch consumed (Scal::used_v / used_g).
// Image buffers: plain iterations alternate llv[0] / llv[1]; a replacement iteration uses 0, 1, 2 | 0, 1, 3 -- a buffer is
// rewritten only after a dot group has been applied in between, i.e. after EVERY workgroup has finished reading it.
// ------------------------------------------------------------------------------------------------------------------
template <int R> struct RowState {
    uint32_t sbase[R], slen[R];           // wave-uniform: first entry / padded length of sub-row j's slice
    uint32_t lens[R];                     // per lane: entries of the row | entries of its diag part << 16
    uint32_t row[R];
    bool live[R];
};

struct WgCtx {
    double *win, *zs, *mval;
    unsigned short *mslot;
    const uint2 *runs;
    unsigned nrw, nrt, nruns, nslots, ns0, ns1, wave, lane;
    bool comm;
    uint32_t row0, nmine, e0;
};

template <int R, bool LDSMAT, bool MULTI>
__device__ __forceinline__ void wg_setup(const PersistArgs &a, unsigned wg, double *dyn, WgCtx &W, RowState<R> &rs)
{
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    W.lane = tid & 63u;
    W.wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    W.win = dyn;
    W.mval = dyn + a.win_slots;
    W.mslot = reinterpret_cast<unsigned short *>(W.mval + a.mat_entries);
    uint2 *runs = reinterpret_cast<uint2 *>(W.mslot + ((a.mat_entries + 3u) & ~3u));
    W.runs = runs;
    W.zs = reinterpret_cast<double *>(runs + a.max_runs);          // [64 * spw * R] the workgroup's values of the vector being published
    W.nrw = a.spw; W.nrt = 64u * a.spw;
    W.comm = W.wave == W.nrw;
    const uint32_t per = a.spw * (uint32_t)R;                       // slices per workgroup
    const uint32_t s0 = wg * per, s1 = min(a.nslices, s0 + per);
    W.row0 = s0 * kSliceRows; W.nmine = min(a.nrows, s1 * kSliceRows) - W.row0;
    W.e0 = a.pbase[s0];
    const uint32_t e1 = a.pbase[s1];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t slice = s0 + (uint32_t)j * W.nrw + W.wave;
        const bool have = !W.comm && slice < s1;
        rs.sbase[j] = 0u; rs.slen[j] = 0u;
        if (have) { rs.sbase[j] = a.pbase[slice]; rs.slen[j] = (a.pbase[slice + 1] - rs.sbase[j]) / kSliceRows; }
        rs.row[j] = slice * kSliceRows + W.lane;
        rs.live[j] = have && rs.row[j] < a.nrows;
        rs.lens[j] = rs.live[j] ? ((uint32_t)a.rlen[rs.row[j]] | ((uint32_t)a.rdiag[rs.row[j]] << 16)) : 0u;
    }
    const unsigned r0w = a.win_ptr[wg];
    W.nruns = a.win_ptr[wg + 1] - r0w;
    for (unsigned i = tid; i < W.nruns; i += nt) runs[i] = a.win_runs[r0w + i];
    W.nslots = 0;
    if (W.nruns) { const uint2 last = a.win_runs[r0w + W.nruns - 1]; W.nslots = (last.y >> 16) + (last.y & 0xFFFFu); }
    if (LDSMAT)
        for (uint32_t j = W.e0 + tid; j < e1; j += nt) { W.mval[j - W.e0] = a.pval[j]; W.mslot[j - W.e0] = a.pslot[j]; }
    W.ns0 = MULTI ? a.snd_ptr[wg] : 0u; W.ns1 = MULTI ? a.snd_ptr[wg + 1] : 0u;
}

// One hand-off and the product that consumes it: every row publishes val (LL image `buf`, tag vseq0 + nv; this workgroup's
// own copy in zs), optionally together with the N dot partials of group g (table and scalar row g & 1); the communication
// wavefront sends partials and halo values and -- wait_scal -- fetches the group's applied scalars into L.sc while the row
// wavefronts stage their window and multiply. out = A val on the thread's rows. Ends with a barrier: L.sc / L.fail are
// valid for everybody.
struct NoMid { __device__ __forceinline__ void operator()() const {} };
// MID: work of the row wavefronts that needs nothing from the hand-off (the shifted kernel's pass over the other shifts' vectors):
// it runs after the values and partials have left -- the neighbours' values and the helper's chain travel meanwhile -- and before
// the window is staged.
template <int R, int N, bool LDSMAT, bool MULTI, bool COEF = false, class MID = NoMid>
__device__ __forceinline__ void xprod(const PersistArgs &a, const WgCtx &W, const RowState<R> &rs, PersistLds &L, const double (&val)[R],
                                      double (&acc)[N > 0 ? N : 1], unsigned buf, unsigned nv, unsigned g, bool wait_scal, double (&out)[R],
                                      MID mid = MID())
{
    const unsigned vtag = a.vseq0 + nv, htag = a.halo_seq0 + nv, gtag = a.seq0 + g;
    llword *const img = a.llv[buf];
    if (!W.comm) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (rs.live[j]) ll_store16_agent(img + 2 * (size_t)rs.row[j], val[j], vtag);
            W.zs[(unsigned)j * W.nrt + threadIdx.x] = val[j];
        }
        if (N > 0) {
#pragma unroll
            for (int d = 0; d < N; ++d) acc[d] = wave_sum(acc[d]);
            if (W.lane == 0) {
#pragma unroll
                for (int d = 0; d < N; ++d) L.wsum[W.wave * kMaxDots + d] = acc[d];
            }
        }
    }
    lds_barrier();
    if (W.comm) {
        // the dot partials first: the helper's chain (table -> sums -> recurrence -> scalars back) is the longest of the phase
        if (N > 0) comm_partials<(N > 0 ? N : 1)>(W.lane, W.nrw, a.dtab[g & 1u] + (size_t)blockIdxWg(a) * kRedSlots * 2, gtag, L);
        comm_halo<MULTI>(a, W.lane, W.zs, htag, W.ns0, W.ns1);
    } else {
        mid();
        const unsigned wgw = blockIdxWg(a);
        const bool logw = MULTI && !COEF && a.waitlog && threadIdx.x == 0 && (wgw == 0u || wgw + 1u == a.nwg);   // (not in the shifted kernels: they sit at their register limit)
        const unsigned long long t_win = logw ? wall_clock64() : 0ull;
        stage_window<MULTI>(a, W.runs, W.nruns, W.nslots, img, vtag, htag, W.win, W.nrt, L, W.zs, W.row0, W.nmine);
        if (logw) a.waitlog[(wgw == 0u ? 1u : 2u) * a.waitcap + (nv - 1u) % a.waitcap] = (unsigned)(wall_clock64() - t_win) | 1u;
    }
    lds_barrier();
    if (W.comm) {
        if (wait_scal) comm_scalars(a, W.lane, a.arow[g & 1u], gtag, L);
        if (COEF && wait_scal) comm_coefs(a, W.lane, a.crow[g & 1u], gtag, L);
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const double *mv = LDSMAT ? W.mval + (rs.sbase[j] - W.e0) : a.pval + rs.sbase[j];
            const unsigned short *ms = LDSMAT ? W.mslot + (rs.sbase[j] - W.e0) : a.pslot + rs.sbase[j];
            // eight rows per thread: 16 entries of a row in flight -- a Transport row in ONE round trip; a product is a chain
            // of R x (batches per row) dependent round trips
            out[j] = persist_row<MULTI, (R > 2 ? 16 : 8)>(mv, ms, rs.slen[j], rs.lens[j] & 0xFFFFu, rs.lens[j] >> 16, W.win);
            if (a.has_shift && rs.live[j]) out[j] += a.shift * val[j];         // (A + sigma I) x, as sell_row does     (src/shifted_solver.c:260)
        }
    }
    lds_barrier();
}

// dot partials that exist only AFTER a product (the drift check): published and answered at once
template <int N>
__device__ __forceinline__ void group_now(const PersistArgs &a, const WgCtx &W, PersistLds &L, double (&acc)[N], unsigned g)
{
    if (!W.comm) {
#pragma unroll
        for (int d = 0; d < N; ++d) acc[d] = wave_sum(acc[d]);
        if (W.lane == 0) {
#pragma unroll
            for (int d = 0; d < N; ++d) L.wsum[W.wave * kMaxDots + d] = acc[d];
        }
    }
    lds_barrier();
    if (W.comm) {
        comm_partials<N>(W.lane, W.nrw, a.dtab[g & 1u] + (size_t)blockIdxWg(a) * kRedSlots * 2, a.seq0 + g, L);
        comm_scalars(a, W.lane, a.arow[g & 1u], a.seq0 + g, L);
    }
    lds_barrier();
}

// the reference's schedule (src/solver.c:498, 522) and the host's request for the first iteration of the launch
__device__ __forceinline__ bool replaces_fixed(const PersistArgs &a, int it)
{
    const int k = a.it0 + it;
    return (it == 0 && a.force_first) || (a.krr > 0 && k > 0 && k % a.krr == 0 && k <= a.krr * a.nrr);
}
__device__ __forceinline__ bool drift_due(const PersistArgs &a, int it)
{
    return a.drift_every > 0 && a.it0 + it > 0 && it % a.drift_every == 0;
}

template <int R, bool LDSMAT, bool MULTI>
__global__ void __launch_bounds__(R <= 2 ? 1024 : 512) __attribute__((amdgpu_waves_per_eu(R <= 2 ? 4 : 2, R <= 2 ? 4 : 2)))
k_pipe_persist(PersistArgs a)
{
    extern __shared__ double dyn[];
    __shared__ PersistLds L;
    const unsigned tid = threadIdx.x;
    const unsigned wg = persist_wg(a);
    if (tid == 0) { L.priv = *a.S; L.fail = 0; L.drift_flag = 0; L.adaptive = 0; }
    lds_barrier();

    if (wg == a.nwg) {
        // ---------------- helper: no rows. Turns dot partials into applied scalars: [drift group,] omega group, end group
        unsigned g = 0, nv = 0;
        for (int it = 0; it < a.niter; ++it) {
            if (L.priv.done) break;
            bool replace = replaces_fixed(a, it);
            if (drift_due(a, it)) {
                ++g; ++nv;
                if (!helper_group<2>(a, a.dtab[g & 1u], a.arow[g & 1u], a.seq0 + g, a.p2p.seq + g - 1u, PH_DRIFT, L, nullptr)) break;
                replace = replace || L.drift_flag != 0;
            }
            unsigned long long *st = a.dbg && it < 32 ? a.dbg + it * 32 + 16 : nullptr;
            ++g;
            if (!helper_group<2>(a, a.dtab[g & 1u], a.arow[g & 1u], a.seq0 + g, a.p2p.seq + g - 1u, PH_OMEGA, L, st ? st + 12 : nullptr)) break;
            if (a.dbg && tid == 0 && it < 32) a.dbg[it * 32 + 10] = wall_clock64();
            ++g;
            if (!helper_group<5>(a, a.dtab[g & 1u], a.arow[g & 1u], a.seq0 + g, a.p2p.seq + g - 1u, PH_RECUR_END, L, nullptr)) break;
            if (a.dbg && tid == 0 && it < 32) a.dbg[it * 32 + 11] = wall_clock64();
            nv += replace ? 6u : 2u;
        }
        helper_finish(a, L, (double)nv, (double)g);
        return;
    }

    // ---------------- row workgroup: spw row wavefronts (lane = row, R rows per thread) + one communication wavefront
    WgCtx W;
    RowState<R> rs;
    wg_setup<R, LDSMAT, MULTI>(a, wg, dyn, W, rs);
    const bool comm = W.comm;
    const Vecs &e = a.v;
    // x and r# are touched once per iteration: with R >= 4 rows per thread they stay in memory (the rank's vectors sit in the
    // Infinity Cache) and are read / written where phase 2 needs them; y = w - alpha z is formed again in phase 2 from the
    // same operands (same bits) instead of being carried across the product.
    constexpr bool XMEM = R >= 4;
    constexpr bool RRP = R == 1;          // replacement iterations and the drift check: one row per thread only (latency-bound ranks)
    constexpr int RX = XMEM ? 1 : R;
    double xr[RX], hr[RX];
    constexpr bool PMEM = R >= 7;         // ... and so does p (read and written in phase 1, read in phase 2)
    constexpr int RP = PMEM ? 1 : R;
    double pr[RP];
    double r[R], s[R], z[R], w[R], v[R], t[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t q = rs.live[j] ? rs.row[j] : 0u;
        if (!XMEM) { xr[j % RX] = e.x[q]; hr[j % RX] = e.rh[q]; }
        if (!PMEM) pr[j % RP] = e.p[q];
        r[j] = e.r[q]; s[j] = e.s[q]; z[j] = e.z[q]; w[j] = e.w[q]; v[j] = e.v[q]; t[j] = e.t[q];
    }
    auto ldp = [&](int j) -> double { return PMEM ? (rs.live[j] ? e.p[rs.row[j]] : 0.0) : pr[j % RP]; };
    auto stp = [&](int j, double val) { if (PMEM) { if (rs.live[j]) e.p[rs.row[j]] = val; } else pr[j % RP] = val; };
    auto ldx = [&](int j) -> double { return XMEM ? (rs.live[j] ? e.x[rs.row[j]] : 0.0) : xr[j % RX]; };
    auto ldh = [&](int j) -> double { return XMEM ? (rs.live[j] ? e.rh[rs.row[j]] : 0.0) : hr[j % RX]; };
    auto stx = [&](int j, double val) { if (XMEM) { if (rs.live[j]) e.x[rs.row[j]] = val; } else xr[j % RX] = val; };
    double alpha = L.priv.alpha, beta = L.priv.beta, omega = L.priv.omega;
    int done = L.priv.done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    const bool trace = a.dbg && wg == a.nwg / 2 && (tid == 0 || tid == W.nrt);
    const int tr0 = tid == 0 ? 0 : 32;
#define STAMP(i) do { if (trace && it < 32) a.dbg[(it * 2 + (tr0 ? 1 : 0)) * 16 + (i)] = wall_clock64(); } while (0)
    unsigned g = 0, nv = 0;
    double none[1] = {0.0};
    for (int it = 0; it < a.niter && !done; ++it) {
        bool replace = RRP && replaces_fixed(a, it);
        if (RRP && drift_due(a, it)) {
            // ---- adaptive replacement: ||(b - A x) - r||^2 against ||r||^2 (FDrift); the helper decides
            double ax[R], xv_[R];
#pragma unroll
            for (int j = 0; j < R; ++j) xv_[j] = comm ? 0.0 : ldx(j);
            xprod<R, 0, LDSMAT, MULTI>(a, W, rs, L, xv_, none, 0u, ++nv, 0u, false, ax);
            if (L.fail) break;
            double acc[2] = {0.0, 0.0};
            if (!comm) {
#pragma unroll
                for (int j = 0; j < R; ++j)
                    if (rs.live[j]) {
                        const double dlt = (e.b[rs.row[j]] + (-1.0) * ax[j]) - r[j];
                        acc[0] += dlt * dlt; acc[1] += r[j] * r[j];
                    }
            }
            group_now<2>(a, W, L, acc, ++g);
            if (L.fail) break;
            replace = replace || L.sc[3] == 2.0;
        }
        STAMP(0);
        if (!RRP || !replace) {
            // ---- phase 1: p, s, z recurrences, q (kept in r), y, (q,y), (y,y)            (src/solver.c:352-364, FPipe1)
            double acc2[2] = {0.0, 0.0};
            if (!comm) {
                double pin[R];
#pragma unroll
                for (int j = 0; j < R; ++j) pin[j] = ldp(j);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    stp(j, recur3<double>(pin[j], s[j], r[j], omega, beta));
                    const double s1n = recur3<double>(s[j], z[j], w[j], omega, beta);
                    const double z1n = recur3<double>(z[j], v[j], t[j], omega, beta);
                    s[j] = s1n; z[j] = z1n;
                    r[j] = r[j] + (-alpha) * s[j];             // q
                    const double y = w[j] + (-alpha) * z[j];
                    if (rs.live[j]) { acc2[0] += r[j] * y; acc2[1] += y * y; }
                }
            }
            STAMP(1);
            // ---- v = A z  ||  publish z, halo, partials; wait for omega                  (src/solver.c:363-369)
            xprod<R, 2, LDSMAT, MULTI>(a, W, rs, L, z, acc2, 0u, ++nv, ++g, true, v);
            STAMP(3);
            if (L.fail) break;
            omega = L.sc[2];
            STAMP(4);
            // ---- phase 2: x, r, w, five dots                                             (src/solver.c:370-380, FPipe2)
            double acc5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            if (!comm) {
                double xin[R], hin[R], pin[R];
#pragma unroll
                for (int j = 0; j < R; ++j) { xin[j] = ldx(j); hin[j] = ldh(j); pin[j] = ldp(j); }      // (in memory: all loads in flight together)
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const double q = r[j];
                    const double y = w[j] + (-alpha) * z[j];           // phase 1's y, bit for bit (w, z, alpha unchanged since)
                    double xx = xin[j] + alpha * pin[j];
                    xx = xx + omega * q;
                    stx(j, xx);
                    r[j] = q + (-omega) * y;
                    const double tt = t[j] + (-alpha) * v[j];
                    w[j] = y + (-omega) * tt;
                    const double h = hin[j];
                    if (rs.live[j]) { acc5[0] += r[j] * r[j]; acc5[1] += h * r[j]; acc5[2] += h * w[j]; acc5[3] += h * s[j]; acc5[4] += h * z[j]; }
                }
            }
            STAMP(5);
            // ---- t = A w  ||  publish w, halo, partials; wait for beta, alpha, done       (src/solver.c:377-390)
            xprod<R, 5, LDSMAT, MULTI>(a, W, rs, L, w, acc5, 1u, ++nv, ++g, true, t);
            STAMP(7);
            if (L.fail) break;
        } else {
            // ---- residual replacement (src/solver.c:494-548; FPUpdate, FQY, FXUpdate, FTrueRes, FDots5)
            double ax[R], pn[R];
#pragma unroll
            for (int j = 0; j < R; ++j) pn[j] = 0.0;
            if (!comm) {
#pragma unroll
                for (int j = 0; j < R; ++j) pn[j] = recur3<double>(ldp(j), s[j], r[j], omega, beta);
#pragma unroll
                for (int j = 0; j < R; ++j) stp(j, pn[j]);
            }
            xprod<R, 0, LDSMAT, MULTI>(a, W, rs, L, pn, none, 0u, ++nv, 0u, false, s);          // s = A p
            if (L.fail) break;
            xprod<R, 0, LDSMAT, MULTI>(a, W, rs, L, s, none, 1u, ++nv, 0u, false, z);           // z = A s
            if (L.fail) break;
            double acc2[2] = {0.0, 0.0};
            if (!comm) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    r[j] = r[j] + (-alpha) * s[j];             // q
                    w[j] = w[j] + (-alpha) * z[j];             // y (kept in w, as FQY does)
                    if (rs.live[j]) { acc2[0] += r[j] * w[j]; acc2[1] += w[j] * w[j]; }
                }
            }
            xprod<R, 2, LDSMAT, MULTI>(a, W, rs, L, z, acc2, 2u, ++nv, ++g, true, v);           // v = A z || (q,y), (y,y) -> omega
            if (L.fail) break;
            omega = L.sc[2];
            double xn[R];
#pragma unroll
            for (int j = 0; j < R; ++j) xn[j] = 0.0;
            if (!comm) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const double xx = ldx(j) + alpha * pn[j];
                    xn[j] = xx + omega * r[j];
                    stx(j, xn[j]);
                }
            }
            xprod<R, 0, LDSMAT, MULTI>(a, W, rs, L, xn, none, 0u, ++nv, 0u, false, ax);         // A x
            if (L.fail) break;
            if (!comm) {
#pragma unroll
                for (int j = 0; j < R; ++j) r[j] = (rs.live[j] ? e.b[rs.row[j]] : 0.0) + (-1.0) * ax[j];   // r = b - A x
            }
            xprod<R, 0, LDSMAT, MULTI>(a, W, rs, L, r, none, 1u, ++nv, 0u, false, w);           // w = A r
            if (L.fail) break;
            double acc5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            if (!comm) {
#pragma unroll
                for (int j = 0; j < R; ++j)
                    if (rs.live[j]) {
                        const double h = ldh(j);
                        acc5[0] += r[j] * r[j]; acc5[1] += h * r[j]; acc5[2] += h * w[j]; acc5[3] += h * s[j]; acc5[4] += h * z[j];
                    }
            }
            xprod<R, 5, LDSMAT, MULTI>(a, W, rs, L, w, acc5, 3u, ++nv, ++g, true, t);           // t = A w || the five -> beta, alpha, k++
            if (L.fail) break;
        }
        alpha = L.sc[0]; beta = L.sc[1]; omega = L.sc[2]; done = L.sc[3] == 1.0 ? 1 : 0;
        STAMP(8);
    }
#undef STAMP
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (rs.live[j]) {
            const uint32_t q = rs.row[j];
            if (!XMEM) e.x[q] = xr[j % RX];
            if (!PMEM) e.p[q] = pr[j % RP];
            e.r[q] = r[j]; e.s[q] = s[j]; e.z[q] = z[j]; e.w[q] = w[j]; e.v[q] = v[j]; e.t[q] = t[j];
        }
}


// The pass over the other shifts' vectors of one row (reference src/shifted_solver.c:264-269, 296-299 / 806-807, 834-837;
// FShiftUpdate / FShPipe2 of bicg_kernels.hip, operation for operation): p_j and x_j of the row are read once and written once
// with the coefficients the helper published (L.coef). B shifts in flight per thread (2 B loads of 8 bytes). On a
// latency-bound rank the two sets (2 x nsig x rows x 8 bytes: 51 MB for 16 shifts on 200 k rows) live in the Infinity Cache
// between iterations -- ordinary loads and stores; non-temporal ones (set_nt) bypass it and are for sets that do not fit.
__device__ __forceinline__ void shift_pass(const PersistArgs &a, const PersistLds &L, uint32_t row, double q, double ro)
{
    constexpr int B = 8;
    for (int j0 = 0; j0 < a.nsig; j0 += B) {
        double pj[B], xj[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int j = j0 + b;
            const bool on = j < a.nsig && j != a.seed;
            const double *pp = a.pset + (size_t)j * a.set_stride + row, *xp = a.xset + (size_t)j * a.set_stride + row;
            pj[b] = !on ? 0.0 : a.set_nt ? __builtin_nontemporal_load(pp) : *pp;
            xj[b] = !on ? 0.0 : a.set_nt ? __builtin_nontemporal_load(xp) : *xp;
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int j = j0 + b;
            if (j >= a.nsig || j == a.seed) continue;
            const double bj = L.coef[0 * kPersistMaxShifts + j], aj = L.coef[1 * kPersistMaxShifts + j];
            const double cp = L.coef[2 * kPersistMaxShifts + j], cx = L.coef[3 * kPersistMaxShifts + j];
            const double c1 = L.coef[4 * kPersistMaxShifts + j], c2 = L.coef[5 * kPersistMaxShifts + j];
            double pp = bj * pj[b];                  // (:265 / :806)
            pp = pp + cp * ro;                       // (:266 / :807)
            double xv = xj[b] + cx * q;              // (:296 / :834)
            xv = xv + aj * pp;                       // (:297 / :835)
            pp = pp + c1 * q;                        // (:298 / :836)
            pp = pp + c2 * ro;                       // (:299 / :837)
            double *xo = a.xset + (size_t)j * a.set_stride + row, *po = a.pset + (size_t)j * a.set_stride + row;
            if (a.set_nt) { __builtin_nontemporal_store(xv, xo); __builtin_nontemporal_store(pp, po); }
            else { *xo = xv; *po = pp; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// shifted_pipe_lopbicgstab (reference src/shifted_solver.c:794-866; FShPipe1 / FShPipe2 / apply_phase_shifted, operation for
// operation) in the persistent form: the SEED system's recurrence is the pipelined iteration above with products of
// A + sigma_seed I; per iteration the helper also runs the per-shift scalar recurrences (one thread per shift) and sends every
// shift's six coefficients along with omega; phase 2 then streams each other shift's p_j, x_j of the thread's row through
// registers (read once, written once: 32 bytes per shift and row, from the Infinity Cache on a latency-bound rank). One row
// per thread, <= kPersistMaxShifts shifts. Two hand-offs and two groups per iteration, numbered like the pipelined kernel's.
// ------------------------------------------------------------------------------------------------------------------
template <bool LDSMAT, bool MULTI>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_shpipe_persist(PersistArgs a)
{
    extern __shared__ double dyn[];
    __shared__ PersistLds L;
    const unsigned tid = threadIdx.x;
    const unsigned wg = persist_wg(a);
    if (tid == 0) { L.priv = *a.S; L.fail = 0; L.drift_flag = 0; L.adaptive = 0; }
    lds_barrier();

    if (wg == a.nwg) {
        unsigned g = 0, nv = 0;
        for (int it = 0; it < a.niter; ++it) {
            if (L.priv.done) break;
            ++g;
            if (!helper_group<2, true>(a, a.dtab[g & 1u], a.arow[g & 1u], a.seq0 + g, a.p2p.seq + g - 1u, PH_SHP_OMEGA, L, nullptr, dyn, a.crow[g & 1u])) break;
            ++g;
            if (!helper_group<5, true>(a, a.dtab[g & 1u], a.arow[g & 1u], a.seq0 + g, a.p2p.seq + g - 1u, PH_SHP_END, L, nullptr, dyn, nullptr)) break;
            nv += 2u;
        }
        helper_finish(a, L, (double)nv, (double)g);
        return;
    }

    WgCtx W;
    RowState<1> rs;
    wg_setup<1, LDSMAT, MULTI>(a, wg, dyn, W, rs);
    const bool comm = W.comm, live = rs.live[0];
    const uint32_t row = live ? rs.row[0] : 0u;
    const Vecs &e = a.v;
    double x = e.x[row], r[1] = {e.r[row]}, p = e.p[row], s = e.s[row], z[1] = {e.z[row]}, w[1] = {e.w[row]}, v[1] = {e.v[row]}, t[1] = {e.t[row]};
    const double h = e.rh[row];
    double ro = 0.0;
    double alpha = L.priv.alpha, beta = L.priv.beta, omega = L.priv.omega;
    int done = L.priv.done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    unsigned g = 0, nv = 0;
    for (int it = 0; it < a.niter && !done; ++it) {
        // ---- phase 1: p[seed], s, z recurrences ; r_old = r ; q (in r), y (in w) ; (q,y), (y,y)      (:794-813, FShPipe1)
        double acc2[2] = {0.0, 0.0};
        if (!comm) {
            p = recur3<double>(p, s, r[0], omega, beta);
            const double s1 = recur3<double>(s, z[0], w[0], omega, beta);
            const double z1 = recur3<double>(z[0], v[0], t[0], omega, beta);
            s = s1; z[0] = z1;
            ro = r[0];
            r[0] = r[0] + (-alpha) * s;
            w[0] = w[0] + (-alpha) * z[0];
            if (live) { acc2[0] = r[0] * w[0]; acc2[1] = w[0] * w[0]; }
        }
        // ---- v = (A + sigma I) z  ||  omega and every shift's coefficients                          (:814-839)
        xprod<1, 2, LDSMAT, MULTI, true>(a, W, rs, L, z, acc2, 0u, ++nv, ++g, true, v);
        if (L.fail) break;
        omega = L.sc[2];
        // ---- phase 2: x[seed] ; r ; w ; five dots -- then, while w and the partials travel, every p_j, x_j     (:829-848, FShPipe2)
        double acc5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        const double q = r[0];
        if (!comm) {
            const double y = w[0];
            const double xx = x + alpha * p;
            x = xx + omega * q;
            r[0] = q + (-omega) * y;                             // (:840)
            const double tt = t[0] + (-alpha) * v[0];            // (:842)
            w[0] = y + (-omega) * tt;                            // (:843)
            if (live) { acc5[0] = r[0] * r[0]; acc5[1] = h * r[0]; acc5[2] = h * w[0]; acc5[3] = h * s; acc5[4] = h * z[0]; }
        }
        auto shifts = [&]() { if (live) shift_pass(a, L, row, q, ro); };
        // ---- t = (A + sigma I) w  ||  the other shifts  ||  beta, alpha, the stopping test            (:849-866)
        // (round 4, first form: the pass over the shifts BEFORE the hand-off: 31.4 us per iteration for 16 shifts on 200 k rows)
        xprod<1, 5, LDSMAT, MULTI, false>(a, W, rs, L, w, acc5, 1u, ++nv, ++g, true, t, shifts);
        if (L.fail) break;
        alpha = L.sc[0]; beta = L.sc[1]; omega = L.sc[2]; done = L.sc[3] == 1.0 ? 1 : 0;
    }
    if (live) {
        e.x[row] = x; e.r[row] = r[0]; e.p[row] = p; e.s[row] = s; e.z[row] = z[0]; e.w[row] = w[0]; e.v[row] = v[0]; e.t[row] = t[0];
        e.ax[row] = ro;                                          // r_old lives in ax (launch_shift_pipe1)
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Plain BiCGStab (reference src/solver.c:86-120) in the same persistent form. Its three scalars are each needed by the
// very next element-wise step (alpha before q, omega before x / r, beta before p), so the three reductions of an
// iteration are exposed -- partial -> helper -> scalars back, ~3 us each -- where the pipelined recurrence hides its two
// behind the products; what goes away against the five-launch form are the five kernel boundaries and every vector
// access: 39 -> ~20 us per iteration on a 200 k-row rank. Expressions: FPlainQ / FPlainXR / FPlainP, operation for operation.
// Groups per iteration (tags seq0 + 3 it + 1 / 2 / 3): (r#,s) -> alpha ; (q,y),(y,y) -> omega ; (r,r),(r#,r) -> beta, k++.
// ------------------------------------------------------------------------------------------------------------------
template <bool LDSMAT, bool MULTI>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_plain_persist(PersistArgs a)
{
    extern __shared__ double dyn[];
    __shared__ PersistLds L;
    const unsigned tid = threadIdx.x, nt = blockDim.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned wg = persist_wg(a);
    if (tid == 0) { L.priv = *a.S; L.fail = 0; L.drift_flag = 0; L.adaptive = 0; }
    lds_barrier();

    if (wg == a.nwg) {
        for (int it = 0; it < a.niter; ++it) {
            if (L.priv.done) break;
            const unsigned s1 = a.seq0 + 3u * (unsigned)it + 1u, m1 = a.p2p.seq + 3u * (unsigned)it;
            if (!helper_group<1>(a, a.dtab[0], a.arow[0], s1, m1, PH_PLAIN_ALPHA, L, nullptr)) break;
            if (!helper_group<2>(a, a.dtab[1], a.arow[1], s1 + 1u, m1 + 1u, PH_OMEGA, L, nullptr)) break;
            if (!helper_group<2>(a, a.dtab[0], a.arow[0], s1 + 2u, m1 + 2u, PH_PLAIN_END, L, nullptr)) break;
        }
        helper_finish(a, L);
        return;
    }

    const RowWg R = row_setup<LDSMAT, MULTI>(a, wg, dyn);
    double *const win = R.win, *const zs = R.zs;
    const uint2 *const runs = R.runs;
    const unsigned nrw = R.nrw, nrt = R.nrt, nruns = R.nruns, nslots = R.nslots, ns0 = R.ns0, ns1 = R.ns1;
    const bool comm = R.comm, live = R.live;
    const uint32_t row0 = R.row0, nmine = R.nmine, row = R.row, slen = R.slen, mylen = R.mylen, mydiag = R.mydiag;
    const double *const gval = R.gval;
    const unsigned short *const gslot = R.gslot;
    const Vecs &e = a.v;
    const uint32_t rr_ = live ? row : 0u;
    double x = e.x[rr_], r = e.r[rr_], p = e.p[rr_], s = e.s[rr_], y = 0.0, q = 0.0;
    const double h = e.rh[rr_];
    double alpha = L.priv.alpha, beta = L.priv.beta, omega = L.priv.omega;
    int done = L.priv.done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    llword *const tab0 = R.tab0, *const tab1 = R.tab1;
    llword *const img0 = R.img0, *const img1 = R.img1;
    for (int it = 0; it < a.niter && !done; ++it) {
        const unsigned g1 = a.seq0 + 3u * (unsigned)it + 1u, g2 = g1 + 1u, g3 = g1 + 2u;
        const unsigned hp = a.halo_seq0 + 2u * (unsigned)it + 1u, hq = hp + 1u;
        // ---- s = A p ; (r#,s) -> alpha                                               (src/solver.c:88-93)
        if (!comm) { publish_only(p, zs, img0, g1); }
        lds_barrier();
        if (comm) comm_halo<MULTI>(a, lane, zs, hp, ns0, ns1);
        else stage_window<MULTI>(a, runs, nruns, nslots, a.llv[0], g1, hp, win, nrt, L, zs, row0, nmine);
        lds_barrier();
        if (!comm) {
            s = persist_row<MULTI>(gval, gslot, slen, mylen, mydiag, win);
            double acc[1] = {live ? h * s : 0.0};
            hand_over<1>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<1>(lane, nrw, tab0, g1, L); comm_scalars(a, lane, a.arow[0], g1, L); }
        lds_barrier();
        if (L.fail) break;
        alpha = L.sc[0];
        // ---- q = r - alpha s ; y = A q ; (q,y), (y,y) -> omega                       (src/solver.c:94-104)
        if (!comm) { q = r + (-alpha) * s; publish_only(q, zs, img1, g2); }
        lds_barrier();
        if (comm) comm_halo<MULTI>(a, lane, zs, hq, ns0, ns1);
        else stage_window<MULTI>(a, runs, nruns, nslots, a.llv[1], g2, hq, win, nrt, L, zs, row0, nmine);
        lds_barrier();
        if (!comm) {
            y = persist_row<MULTI>(gval, gslot, slen, mylen, mydiag, win);
            double acc[2] = {live ? q * y : 0.0, live ? y * y : 0.0};
            hand_over<2>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<2>(lane, nrw, tab1, g2, L); comm_scalars(a, lane, a.arow[1], g2, L); }
        lds_barrier();
        if (L.fail) break;
        omega = L.sc[2];
        // ---- x += alpha p + omega q ; r = q - omega y ; (r,r), (r#,r) -> beta, k++   (src/solver.c:105-116)
        if (!comm) {
            double xx = x + alpha * p;
            xx = xx + omega * q;
            x = xx;
            r = q + (-omega) * y;
            double acc[2] = {live ? r * r : 0.0, live ? h * r : 0.0};
            hand_over<2>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<2>(lane, nrw, tab0, g3, L); comm_scalars(a, lane, a.arow[0], g3, L); }
        lds_barrier();
        if (L.fail) break;
        beta = L.sc[1]; done = L.sc[3] != 0.0 ? 1 : 0;
        // ---- p = beta p ; p += r ; p += (-beta omega) s                              (src/solver.c:117-119)
        if (!comm && !done) {
            double pp = beta * p;
            pp = pp + 1.0 * r;
            pp = pp + (-beta * omega) * s;
            p = pp;
        }
        lds_barrier();                        // L.sc is rewritten by the next group
    }
    if (live) { e.x[row] = x; e.r[row] = r; e.p[row] = p; e.s[row] = s; e.y[row] = y; }
}


// Plain BiCGStab with R rows per thread (round 6: ranks of 4 GPUs -- 400 k rows of Transport = 1 565 rows per CU -- ran the
// five-launch iteration at 46.5 us where the pipelined method's persistent kernel took 25.1). The protocol, tags, tables and the
// helper are k_plain_persist's, operation for operation; a thread keeps x, r, p, s, y, q, r# of its R rows in registers (thread
// (wavefront w, lane l) owns rows (s0 + j nrw + w) 64 + l, position j nrt + tid of the workgroup's range, like k_pipe_persist),
// the matrix slices are streamed from memory. Dot partials: a thread adds its rows' products in row order before the wavefront
// sum -- an association of its own, like every tiling's.
template <int R, bool MULTI>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_plain_persist_r(PersistArgs a)
{
    extern __shared__ double dyn[];
    __shared__ PersistLds L;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wg = persist_wg(a);
    if (tid == 0) { L.priv = *a.S; L.fail = 0; L.drift_flag = 0; L.adaptive = 0; }
    lds_barrier();

    if (wg == a.nwg) {
        for (int it = 0; it < a.niter; ++it) {
            if (L.priv.done) break;
            const unsigned s1 = a.seq0 + 3u * (unsigned)it + 1u, m1 = a.p2p.seq + 3u * (unsigned)it;
            if (!helper_group<1>(a, a.dtab[0], a.arow[0], s1, m1, PH_PLAIN_ALPHA, L, nullptr)) break;
            if (!helper_group<2>(a, a.dtab[1], a.arow[1], s1 + 1u, m1 + 1u, PH_OMEGA, L, nullptr)) break;
            if (!helper_group<2>(a, a.dtab[0], a.arow[0], s1 + 2u, m1 + 2u, PH_PLAIN_END, L, nullptr)) break;
        }
        helper_finish(a, L);
        return;
    }

    WgCtx W;
    RowState<R> rs;
    wg_setup<R, false, MULTI>(a, wg, dyn, W, rs);
    const bool comm = W.comm;
    const Vecs &e = a.v;
    double x[R], r[R], p[R], s[R], y[R], q[R], h[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t at = rs.live[j] ? rs.row[j] : 0u;
        x[j] = e.x[at]; r[j] = e.r[at]; p[j] = e.p[at]; s[j] = e.s[at]; h[j] = e.rh[at];
        y[j] = 0.0; q[j] = 0.0;
    }
    double alpha = L.priv.alpha, beta = L.priv.beta, omega = L.priv.omega;
    int done = L.priv.done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    llword *const tab0 = a.dtab[0] + (size_t)wg * kRedSlots * 2, *const tab1 = a.dtab[1] + (size_t)wg * kRedSlots * 2;
    // every row publishes val (LL image buf, tag seq; the workgroup's own copy in zs), then the product out = A val
    auto product = [&](const double (&val)[R], unsigned buf, unsigned seq, unsigned hseq, double (&out)[R]) {
        if (!comm) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (rs.live[j]) ll_store16_agent(a.llv[buf] + 2 * (size_t)rs.row[j], val[j], seq);
                W.zs[(unsigned)j * W.nrt + tid] = val[j];
            }
        }
        lds_barrier();
        if (comm) comm_halo<MULTI>(a, lane, W.zs, hseq, W.ns0, W.ns1);
        else stage_window<MULTI>(a, W.runs, W.nruns, W.nslots, a.llv[buf], seq, hseq, W.win, W.nrt, L, W.zs, W.row0, W.nmine);
        lds_barrier();
        if (!comm) {
#pragma unroll
            for (int j = 0; j < R; ++j)
                out[j] = persist_row<MULTI, 8>(a.pval + rs.sbase[j], a.pslot + rs.sbase[j], rs.slen[j], rs.lens[j] & 0xFFFFu, rs.lens[j] >> 16, W.win);
        }
    };
    for (int it = 0; it < a.niter && !done; ++it) {
        const unsigned g1 = a.seq0 + 3u * (unsigned)it + 1u, g2 = g1 + 1u, g3 = g1 + 2u;
        const unsigned hp = a.halo_seq0 + 2u * (unsigned)it + 1u, hq = hp + 1u;
        // ---- s = A p ; (r#,s) -> alpha                                               (src/solver.c:88-93)
        product(p, 0u, g1, hp, s);
        if (!comm) {
            double acc[1] = {0.0};
#pragma unroll
            for (int j = 0; j < R; ++j) acc[0] += rs.live[j] ? h[j] * s[j] : 0.0;
            hand_over<1>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<1>(lane, W.nrw, tab0, g1, L); comm_scalars(a, lane, a.arow[0], g1, L); }
        lds_barrier();
        if (L.fail) break;
        alpha = L.sc[0];
        // ---- q = r - alpha s ; y = A q ; (q,y), (y,y) -> omega                       (src/solver.c:94-104)
        if (!comm) {
#pragma unroll
            for (int j = 0; j < R; ++j) q[j] = r[j] + (-alpha) * s[j];
        }
        product(q, 1u, g2, hq, y);
        if (!comm) {
            double acc[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < R; ++j) { acc[0] += rs.live[j] ? q[j] * y[j] : 0.0; acc[1] += rs.live[j] ? y[j] * y[j] : 0.0; }
            hand_over<2>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<2>(lane, W.nrw, tab1, g2, L); comm_scalars(a, lane, a.arow[1], g2, L); }
        lds_barrier();
        if (L.fail) break;
        omega = L.sc[2];
        // ---- x += alpha p + omega q ; r = q - omega y ; (r,r), (r#,r) -> beta, k++   (src/solver.c:105-116)
        if (!comm) {
            double acc[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < R; ++j) {
                double xx = x[j] + alpha * p[j];
                xx = xx + omega * q[j];
                x[j] = xx;
                r[j] = q[j] + (-omega) * y[j];
                acc[0] += rs.live[j] ? r[j] * r[j] : 0.0;
                acc[1] += rs.live[j] ? h[j] * r[j] : 0.0;
            }
            hand_over<2>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<2>(lane, W.nrw, tab0, g3, L); comm_scalars(a, lane, a.arow[0], g3, L); }
        lds_barrier();
        if (L.fail) break;
        beta = L.sc[1]; done = L.sc[3] != 0.0 ? 1 : 0;
        // ---- p = beta p ; p += r ; p += (-beta omega) s                              (src/solver.c:117-119)
        if (!comm && !done) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                double pp = beta * p[j];
                pp = pp + 1.0 * r[j];
                pp = pp + (-beta * omega) * s[j];
                p[j] = pp;
            }
        }
        lds_barrier();                        // L.sc is rewritten by the next group
    }
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (rs.live[j]) { const uint32_t row = rs.row[j]; e.x[row] = x[j]; e.r[row] = r[j]; e.p[row] = p[j]; e.s[row] = s[j]; e.y[row] = y[j]; }
}


// ------------------------------------------------------------------------------------------------------------------
// shifted_lopbicgstab (reference src/shifted_solver.c:257-319) in the persistent form: the SEED system's iteration is plain
// BiCGStab on A + sigma_seed I -- three exposed groups, PH_SH_ALPHA / PH_SH_OMEGA / PH_SH_END, numbered like the plain kernel's
// -- and the helper runs the per-shift scalar recurrences with them (one thread per shift) and publishes every shift's six
// coefficients together with omega. The pass over the other shifts' p_j / x_j (shift_pass) runs while the third group's sums
// travel to the helper and back. Expressions: FShiftQ / FShiftUpdate / FShiftPSeed, operation for operation.
// v.x / v.p are x[seed] / p[seed]; r_old lives in registers (and goes back to v.ax like the multi-launch form leaves it).
// ------------------------------------------------------------------------------------------------------------------
template <bool LDSMAT, bool MULTI>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_shlop_persist(PersistArgs a)
{
    extern __shared__ double dyn[];
    __shared__ PersistLds L;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wg = persist_wg(a);
    if (tid == 0) { L.priv = *a.S; L.fail = 0; L.drift_flag = 0; L.adaptive = 0; }
    lds_barrier();

    if (wg == a.nwg) {
        for (int it = 0; it < a.niter; ++it) {
            if (L.priv.done) break;
            const unsigned s1 = a.seq0 + 3u * (unsigned)it + 1u, m1 = a.p2p.seq + 3u * (unsigned)it;
            if (!helper_group<1, true>(a, a.dtab[0], a.arow[0], s1, m1, PH_SH_ALPHA, L, nullptr, dyn, nullptr)) break;
            if (!helper_group<2, true>(a, a.dtab[1], a.arow[1], s1 + 1u, m1 + 1u, PH_SH_OMEGA, L, nullptr, dyn, a.crow[1])) break;
            if (!helper_group<2, true>(a, a.dtab[0], a.arow[0], s1 + 2u, m1 + 2u, PH_SH_END, L, nullptr, dyn, nullptr)) break;
        }
        helper_finish(a, L);
        return;
    }

    const RowWg R = row_setup<LDSMAT, MULTI>(a, wg, dyn);
    double *const win = R.win, *const zs = R.zs;
    const uint2 *const runs = R.runs;
    const unsigned nrw = R.nrw, nrt = R.nrt, nruns = R.nruns, nslots = R.nslots, ns0 = R.ns0, ns1 = R.ns1;
    const bool comm = R.comm, live = R.live;
    const uint32_t row0 = R.row0, nmine = R.nmine, row = R.row, slen = R.slen, mylen = R.mylen, mydiag = R.mydiag;
    const double *const gval = R.gval;
    const unsigned short *const gslot = R.gslot;
    const Vecs &e = a.v;
    const uint32_t rr_ = live ? row : 0u;
    double x = e.x[rr_], r = e.r[rr_], p = e.p[rr_], s = e.s[rr_], y = e.y[rr_], q = 0.0, ro = e.ax[rr_];
    const double h = e.rh[rr_];
    double alpha = L.priv.alpha, beta = L.priv.beta, omega = L.priv.omega;
    int done = L.priv.done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    llword *const tab0 = R.tab0, *const tab1 = R.tab1;
    llword *const img0 = R.img0, *const img1 = R.img1;
    for (int it = 0; it < a.niter && !done; ++it) {
        const unsigned g1 = a.seq0 + 3u * (unsigned)it + 1u, g2 = g1 + 1u, g3 = g1 + 2u;
        const unsigned hp = a.halo_seq0 + 2u * (unsigned)it + 1u, hq = hp + 1u;
        // ---- s = (A + sigma I) p[seed] ; (r#,s) -> alpha[seed], every beta_j, alpha_j                 (:259-287)
        if (!comm) { publish_only(p, zs, img0, g1); }
        lds_barrier();
        if (comm) comm_halo<MULTI>(a, lane, zs, hp, ns0, ns1);
        else stage_window<MULTI>(a, runs, nruns, nslots, a.llv[0], g1, hp, win, nrt, L, zs, row0, nmine);
        lds_barrier();
        if (!comm) {
            s = persist_row<MULTI>(gval, gslot, slen, mylen, mydiag, win);
            if (a.has_shift && live) s += a.shift * p;                       // (:260)
            double acc[1] = {live ? h * s : 0.0};
            hand_over<1>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<1>(lane, nrw, tab0, g1, L); comm_scalars(a, lane, a.arow[0], g1, L); }
        lds_barrier();
        if (L.fail) break;
        alpha = L.sc[0];
        // ---- r_old = r ; q = r - alpha s ; y = (A + sigma I) q ; (q,y), (q,q) -> omega and the coefficients   (:269-300)
        if (!comm) { ro = r; q = r + (-alpha) * s; publish_only(q, zs, img1, g2); }
        lds_barrier();
        if (comm) comm_halo<MULTI>(a, lane, zs, hq, ns0, ns1);
        else stage_window<MULTI>(a, runs, nruns, nslots, a.llv[1], g2, hq, win, nrt, L, zs, row0, nmine);
        lds_barrier();
        if (!comm) {
            y = persist_row<MULTI>(gval, gslot, slen, mylen, mydiag, win);
            if (a.has_shift && live) y += a.shift * q;                       // (:278)
            double acc[2] = {live ? q * y : 0.0, live ? q * q : 0.0};
            hand_over<2>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) {
            comm_partials<2>(lane, nrw, tab1, g2, L);
            comm_scalars(a, lane, a.arow[1], g2, L);
            comm_coefs(a, lane, a.crow[1], g2, L);
        }
        lds_barrier();
        if (L.fail) break;
        omega = L.sc[2];
        // ---- x[seed] += alpha p + omega q ; r = q - omega y ; (r,r), (r#,r) -> beta, k++ ; meanwhile every p_j, x_j   (:292-315)
        if (!comm) {
            double xx = x + alpha * p;
            x = xx + omega * q;
            r = q + (-omega) * y;
            double acc[2] = {live ? r * r : 0.0, live ? h * r : 0.0};
            hand_over<2>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<2>(lane, nrw, tab0, g3, L); comm_scalars(a, lane, a.arow[0], g3, L); }
        else if (live) shift_pass(a, L, row, q, ro);
        lds_barrier();
        if (L.fail) break;
        beta = L.sc[1]; done = L.sc[3] != 0.0 ? 1 : 0;
        // ---- p[seed] = beta p ; += r ; += (-beta omega) s                                            (:317-319)
        if (!comm) {
            double pp = beta * p;
            pp = pp + 1.0 * r;
            pp = pp + (-beta * omega) * s;
            p = pp;
        }
        lds_barrier();                        // L.sc is rewritten by the next group
    }
    if (live) { e.x[row] = x; e.r[row] = r; e.p[row] = p; e.s[row] = s; e.y[row] = y; e.ax[row] = ro; }
}


// ------------------------------------------------------------------------------------------------------------------
// CA-BiCGStab (reference src/solver.c:216-251) in the persistent form: z = A s and w = A r per iteration, two groups --
// (q,y),(y,y) -> omega, needed at once by the x / r update, and (r,r),(r#,r),(r#,w),(r#,s),(r#,z) -> beta, alpha, k++ after
// the second product -- both exposed. Expressions: FCaPS / FQY / FCaXR, operation for operation. The helper's schedule is
// the pipelined solver's (PH_OMEGA, PH_RECUR_END).
// ------------------------------------------------------------------------------------------------------------------
template <bool LDSMAT, bool MULTI>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_ca_persist(PersistArgs a)
{
    extern __shared__ double dyn[];
    __shared__ PersistLds L;
    const unsigned tid = threadIdx.x, nt = blockDim.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned wg = persist_wg(a);
    if (tid == 0) { L.priv = *a.S; L.fail = 0; L.drift_flag = 0; L.adaptive = 0; }
    lds_barrier();

    if (wg == a.nwg) {
        for (int it = 0; it < a.niter; ++it) {
            if (L.priv.done) break;
            const unsigned s1 = a.seq0 + 2u * (unsigned)it + 1u, m1 = a.p2p.seq + 2u * (unsigned)it;
            if (!helper_group<2>(a, a.dtab[0], a.arow[0], s1, m1, PH_OMEGA, L, nullptr)) break;
            if (!helper_group<5>(a, a.dtab[1], a.arow[1], s1 + 1u, m1 + 1u, PH_RECUR_END, L, nullptr)) break;
        }
        helper_finish(a, L);
        return;
    }

    const RowWg R = row_setup<LDSMAT, MULTI>(a, wg, dyn);
    double *const win = R.win, *const zs = R.zs;
    const uint2 *const runs = R.runs;
    const unsigned nrw = R.nrw, nrt = R.nrt, nruns = R.nruns, nslots = R.nslots, ns0 = R.ns0, ns1 = R.ns1;
    const bool comm = R.comm, live = R.live;
    const uint32_t row0 = R.row0, nmine = R.nmine, row = R.row, slen = R.slen, mylen = R.mylen, mydiag = R.mydiag;
    const double *const gval = R.gval;
    const unsigned short *const gslot = R.gslot;
    const Vecs &e = a.v;
    const uint32_t rr_ = live ? row : 0u;
    double x = e.x[rr_], r = e.r[rr_], p = e.p[rr_], s = e.s[rr_], z = e.z[rr_], w = e.w[rr_];
    const double h = e.rh[rr_];
    double alpha = L.priv.alpha, beta = L.priv.beta, omega = L.priv.omega;
    int done = L.priv.done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    llword *const tab0 = R.tab0, *const tab1 = R.tab1;
    llword *const img0 = R.img0, *const img1 = R.img1;
    for (int it = 0; it < a.niter && !done; ++it) {
        const unsigned g1 = a.seq0 + 2u * (unsigned)it + 1u, g2 = g1 + 1u;
        const unsigned hs = a.halo_seq0 + 2u * (unsigned)it + 1u, hr = hs + 1u;
        // ---- p = r + beta (p - omega s) ; s = w + beta (s - omega z) ; z = A s      (src/solver.c:217-224)
        if (!comm) {
            p = recur3<double>(p, s, r, omega, beta);
            s = recur3<double>(s, z, w, omega, beta);
            publish_only(s, zs, img0, g1);
        }
        lds_barrier();
        if (comm) comm_halo<MULTI>(a, lane, zs, hs, ns0, ns1);
        else stage_window<MULTI>(a, runs, nruns, nslots, a.llv[0], g1, hs, win, nrt, L, zs, row0, nmine);
        lds_barrier();
        // ---- q = r - alpha s (in r) ; y = w - alpha z (in w) ; (q,y), (y,y) -> omega   (src/solver.c:225-232)
        if (!comm) {
            z = persist_row<MULTI>(gval, gslot, slen, mylen, mydiag, win);
            r = r + (-alpha) * s;
            w = w + (-alpha) * z;
            double acc[2] = {live ? r * w : 0.0, live ? w * w : 0.0};
            hand_over<2>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<2>(lane, nrw, tab0, g1, L); comm_scalars(a, lane, a.arow[0], g1, L); }
        lds_barrier();
        if (L.fail) break;
        omega = L.sc[2];
        // ---- x += alpha p + omega q ; r = q - omega y ; w = A r                       (src/solver.c:233-239)
        if (!comm) {
            double xx = x + alpha * p;
            xx = xx + omega * r;
            x = xx;
            r = r + (-omega) * w;
            publish_only(r, zs, img1, g2);
        }
        lds_barrier();
        if (comm) comm_halo<MULTI>(a, lane, zs, hr, ns0, ns1);
        else stage_window<MULTI>(a, runs, nruns, nslots, a.llv[1], g2, hr, win, nrt, L, zs, row0, nmine);
        lds_barrier();
        // ---- (r,r), (r#,r), (r#,w), (r#,s), (r#,z) -> beta, alpha, k++                 (src/solver.c:240-251)
        if (!comm) {
            w = persist_row<MULTI>(gval, gslot, slen, mylen, mydiag, win);
            double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            if (live) { acc[0] = r * r; acc[1] = h * r; acc[2] = h * w; acc[3] = h * s; acc[4] = h * z; }
            hand_over<5>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<5>(lane, nrw, tab1, g2, L); comm_scalars(a, lane, a.arow[1], g2, L); }
        lds_barrier();
        if (L.fail) break;
        alpha = L.sc[0]; beta = L.sc[1]; omega = L.sc[2]; done = L.sc[3] != 0.0 ? 1 : 0;
        lds_barrier();                        // L.sc is rewritten by the next group
    }
    if (live) { e.x[row] = x; e.r[row] = r; e.p[row] = p; e.s[row] = s; e.z[row] = z; e.w[row] = w; }
}


// CA-BiCGStab with R rows per thread: k_ca_persist's protocol, tags and helper; rows and registers as in k_plain_persist_r.
template <int R, bool MULTI>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_ca_persist_r(PersistArgs a)
{
    extern __shared__ double dyn[];
    __shared__ PersistLds L;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wg = persist_wg(a);
    if (tid == 0) { L.priv = *a.S; L.fail = 0; L.drift_flag = 0; L.adaptive = 0; }
    lds_barrier();

    if (wg == a.nwg) {
        for (int it = 0; it < a.niter; ++it) {
            if (L.priv.done) break;
            const unsigned s1 = a.seq0 + 2u * (unsigned)it + 1u, m1 = a.p2p.seq + 2u * (unsigned)it;
            if (!helper_group<2>(a, a.dtab[0], a.arow[0], s1, m1, PH_OMEGA, L, nullptr)) break;
            if (!helper_group<5>(a, a.dtab[1], a.arow[1], s1 + 1u, m1 + 1u, PH_RECUR_END, L, nullptr)) break;
        }
        helper_finish(a, L);
        return;
    }

    WgCtx W;
    RowState<R> rs;
    wg_setup<R, false, MULTI>(a, wg, dyn, W, rs);
    const bool comm = W.comm;
    const Vecs &e = a.v;
    double x[R], r[R], p[R], s[R], z[R], w[R], h[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t at = rs.live[j] ? rs.row[j] : 0u;
        x[j] = e.x[at]; r[j] = e.r[at]; p[j] = e.p[at]; s[j] = e.s[at]; z[j] = e.z[at]; w[j] = e.w[at]; h[j] = e.rh[at];
    }
    double alpha = L.priv.alpha, beta = L.priv.beta, omega = L.priv.omega;
    int done = L.priv.done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    llword *const tab0 = a.dtab[0] + (size_t)wg * kRedSlots * 2, *const tab1 = a.dtab[1] + (size_t)wg * kRedSlots * 2;
    auto product = [&](const double (&val)[R], unsigned buf, unsigned seq, unsigned hseq, double (&out)[R]) {
        if (!comm) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (rs.live[j]) ll_store16_agent(a.llv[buf] + 2 * (size_t)rs.row[j], val[j], seq);
                W.zs[(unsigned)j * W.nrt + tid] = val[j];
            }
        }
        lds_barrier();
        if (comm) comm_halo<MULTI>(a, lane, W.zs, hseq, W.ns0, W.ns1);
        else stage_window<MULTI>(a, W.runs, W.nruns, W.nslots, a.llv[buf], seq, hseq, W.win, W.nrt, L, W.zs, W.row0, W.nmine);
        lds_barrier();
        if (!comm) {
#pragma unroll
            for (int j = 0; j < R; ++j)
                out[j] = persist_row<MULTI, 8>(a.pval + rs.sbase[j], a.pslot + rs.sbase[j], rs.slen[j], rs.lens[j] & 0xFFFFu, rs.lens[j] >> 16, W.win);
        }
    };
    for (int it = 0; it < a.niter && !done; ++it) {
        const unsigned g1 = a.seq0 + 2u * (unsigned)it + 1u, g2 = g1 + 1u;
        const unsigned hs = a.halo_seq0 + 2u * (unsigned)it + 1u, hr = hs + 1u;
        // ---- p = r + beta (p - omega s) ; s = w + beta (s - omega z) ; z = A s      (src/solver.c:217-224)
        if (!comm) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                p[j] = recur3<double>(p[j], s[j], r[j], omega, beta);
                s[j] = recur3<double>(s[j], z[j], w[j], omega, beta);
            }
        }
        product(s, 0u, g1, hs, z);
        // ---- q = r - alpha s (in r) ; y = w - alpha z (in w) ; (q,y), (y,y) -> omega   (src/solver.c:225-232)
        if (!comm) {
            double acc[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < R; ++j) {
                r[j] = r[j] + (-alpha) * s[j];
                w[j] = w[j] + (-alpha) * z[j];
                acc[0] += rs.live[j] ? r[j] * w[j] : 0.0;
                acc[1] += rs.live[j] ? w[j] * w[j] : 0.0;
            }
            hand_over<2>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<2>(lane, W.nrw, tab0, g1, L); comm_scalars(a, lane, a.arow[0], g1, L); }
        lds_barrier();
        if (L.fail) break;
        omega = L.sc[2];
        // ---- x += alpha p + omega q ; r = q - omega y ; w = A r                       (src/solver.c:233-239)
        if (!comm) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                double xx = x[j] + alpha * p[j];
                xx = xx + omega * r[j];
                x[j] = xx;
                r[j] = r[j] + (-omega) * w[j];
            }
        }
        product(r, 1u, g2, hr, w);
        // ---- (r,r), (r#,r), (r#,w), (r#,s), (r#,z) -> beta, alpha, k++                 (src/solver.c:240-251)
        if (!comm) {
            double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < R; ++j)
                if (rs.live[j]) { acc[0] += r[j] * r[j]; acc[1] += h[j] * r[j]; acc[2] += h[j] * w[j]; acc[3] += h[j] * s[j]; acc[4] += h[j] * z[j]; }
            hand_over<5>(0.0, acc, nullptr, L, nullptr, 0u);
        }
        lds_barrier();
        if (comm) { comm_partials<5>(lane, W.nrw, tab1, g2, L); comm_scalars(a, lane, a.arow[1], g2, L); }
        lds_barrier();
        if (L.fail) break;
        alpha = L.sc[0]; beta = L.sc[1]; omega = L.sc[2]; done = L.sc[3] != 0.0 ? 1 : 0;
        lds_barrier();                        // L.sc is rewritten by the next group
    }
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (rs.live[j]) { const uint32_t row = rs.row[j]; e.x[row] = x[j]; e.r[row] = r[j]; e.p[row] = p[j]; e.s[row] = s[j]; e.z[row] = z[j]; e.w[row] = w[j]; }
}

}  // namespace

unsigned persist_lds_bytes(const PersistArgs &a)
{
    const size_t rows = 64u * (size_t)a.spw * (a.rpt ? a.rpt : 1u);       // rows of a workgroup
    return (unsigned)(8u * (size_t)a.win_slots + 8u * (size_t)a.mat_entries + 2u * (((size_t)a.mat_entries + 3u) & ~(size_t)3u) +
                      8u * (size_t)a.max_runs + 8u * rows);
}

// Returns hipSuccess, or why the launch did not happen (the caller falls back to the multi-launch iteration): launch errors
// are ALWAYS looked at -- a persistent launch that silently failed would leave the solve at k = 0 --, and before a kernel's
// first launch on a device the runtime is asked whether nwg + 1 workgroups of this size and LDS footprint can be resident
// together (the workgroups spin-wait on each other: co-residency is a correctness condition, not a performance one).
static hipError_t launch_persist(const PersistArgs &a, hipStream_t st, int method)
{
    // (shifted kernel: the helper workgroup's maximum over the shifts runs in the first 8 KiB of the dynamic part)
    const unsigned lds = method >= 3 ? std::max(persist_lds_bytes(a), 8192u) : persist_lds_bytes(a);
    const dim3 g(a.nwg + 1u), b(64u * (a.spw + 1u));        // + the communication wavefront
    auto go = [&](auto kernel, int slot) -> hipError_t {
        static std::set<std::pair<int, int>> ready;        // (device, instantiation)
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int idx = slot * 4 + (a.mat_entries ? 2 : 0) + (a.multi ? 1 : 0) + 64 * (int)a.rpt;
        if (!ready.count({dev, idx})) {       // (the runtime answers "invalid argument" and launches with > 64 KiB of LDS all the same)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersistMaxLds);
            (void)hipGetLastError();
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)b.x, lds) == hipSuccess &&
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
                if (per_cu < 1 || (long long)per_cu * cus < (long long)g.x) {
                    fprintf(stderr, "bicgstab_hip: persistent kernel: %u workgroups of %u threads with %u bytes of LDS cannot be co-resident "
                                    "(%d per CU x %d CUs)\n", g.x, b.x, lds, per_cu, cus);
                    return hipErrorCooperativeLaunchTooLarge;
                }
            } else {
                (void)hipGetLastError();      // no answer: the launch itself will tell
            }
            ready.insert({dev, idx});
        }
        hipLaunchKernelGGL(kernel, g, b, lds, st, a);
        return hipGetLastError();
    };
    hipError_t err;
    if (method == 1 && a.rpt == 2u && !a.mat_entries)                // plain BiCGStab, two rows per thread (400 k-row ranks)
        return a.multi ? go(k_plain_persist_r<2, true>, 5) : go(k_plain_persist_r<2, false>, 5);
    if (method == 2 && a.rpt == 2u && !a.mat_entries)                // CA-BiCGStab, likewise
        return a.multi ? go(k_ca_persist_r<2, true>, 6) : go(k_ca_persist_r<2, false>, 6);
    if (method != 0 && a.rpt != 1u) return hipErrorInvalidValue;      // several rows per thread otherwise: the pipelined kernel only
    if (method == 0) {
        if (a.rpt == 1u) {
            if (a.mat_entries) { err = a.multi ? go(k_pipe_persist<1, true, true>, 0) : go(k_pipe_persist<1, true, false>, 0); }
            else { err = a.multi ? go(k_pipe_persist<1, false, true>, 0) : go(k_pipe_persist<1, false, false>, 0); }
        } else if (a.mat_entries) {
            return hipErrorInvalidValue;                              // (the matrix of such a rank does not fit LDS)
        } else if (a.rpt == 2u) { err = a.multi ? go(k_pipe_persist<2, false, true>, 0) : go(k_pipe_persist<2, false, false>, 0);
        } else if (a.rpt == 8u) { err = a.multi ? go(k_pipe_persist<8, false, true>, 0) : go(k_pipe_persist<8, false, false>, 0);
        } else return hipErrorInvalidValue;
    } else if (method == 3) {
        if (a.nsig < 1 || a.nsig > kPersistMaxShifts) return hipErrorInvalidValue;
        if (a.mat_entries) { err = a.multi ? go(k_shpipe_persist<true, true>, 3) : go(k_shpipe_persist<true, false>, 3); }
        else { err = a.multi ? go(k_shpipe_persist<false, true>, 3) : go(k_shpipe_persist<false, false>, 3); }
    } else if (method == 4) {
        if (a.nsig < 1 || a.nsig > kPersistMaxShifts) return hipErrorInvalidValue;
        if (a.mat_entries) { err = a.multi ? go(k_shlop_persist<true, true>, 4) : go(k_shlop_persist<true, false>, 4); }
        else { err = a.multi ? go(k_shlop_persist<false, true>, 4) : go(k_shlop_persist<false, false>, 4); }
    } else if (method == 1) {
        if (a.mat_entries) { err = a.multi ? go(k_plain_persist<true, true>, 1) : go(k_plain_persist<true, false>, 1); }
        else { err = a.multi ? go(k_plain_persist<false, true>, 1) : go(k_plain_persist<false, false>, 1); }
    } else {
        if (a.mat_entries) { err = a.multi ? go(k_ca_persist<true, true>, 2) : go(k_ca_persist<true, false>, 2); }
        else { err = a.multi ? go(k_ca_persist<false, true>, 2) : go(k_ca_persist<false, false>, 2); }
    }
    if (err != hipSuccess) fprintf(stderr, "bicgstab_hip: persistent kernel launch failed: %s\n", hipGetErrorString(err));
    return err;
}
hipError_t launch_pipe_persist(const PersistArgs &a, hipStream_t st) { return launch_persist(a, st, 0); }
hipError_t launch_shpipe_persist(const PersistArgs &a, hipStream_t st) { return launch_persist(a, st, 3); }
hipError_t launch_shlop_persist(const PersistArgs &a, hipStream_t st) { return launch_persist(a, st, 4); }
hipError_t launch_plain_persist(const PersistArgs &a, hipStream_t st) { return launch_persist(a, st, 1); }
hipError_t launch_ca_persist(const PersistArgs &a, hipStream_t st) { return launch_persist(a, st, 2); }

void preload_persist_kernels()
{
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k_plain_persist<true, false>));
    (void)hipGetLastError();
}

}  // namespace bicg
