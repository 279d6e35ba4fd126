// bicg_reduce.h -- device-side reductions and scalar recurrences shared by every translation unit that launches kernels
// with dot products (bicg_kernels.hip, bicg_stencil.hip): the workgroup sum, the publication of a workgroup's partial sums
// (arrival tickets / tail finish / one partial per wavefront), the consumer-side finish of a dot group and the phases the
// finishing thread applies. Moved here unchanged from bicg_kernels.hip (round 5) so that new kernels get a file of their own.
#pragma once
#include "bicg_device.h"
#include "bicg_devfn.h"

namespace bicg {

// ---- shifted solvers with stop flags and seed switching (reference src/shifted_switching_solver.c).
// One implementation serves shifted_lopbicg (:20-257, SH_FLAG) and shifted_lopbicg_switching
// (:260-608, SH_SWITCH; _noovlp :611-1016 is an arithmetic twin): the per-shift recurrences are the
// same expressions, written in the archive form of the switching variant -- alpha/beta/omega of
// the seed and pi of every shift are kept per iteration, index kk = completed iterations + 1 (the
// reference's k of the switching variant), index 0 = the initial values.
__device__ __forceinline__ void apply_phase_switching(Scal *S, int phase)
{
    ShiftDev *H = S->sh;
    const double *d = S->red;
    const int nsig = H->nsig, L = H->arc_len;
    const int kk = S->k + 1;
    const int tid = threadIdx.x;
#define PI_(j, i) H->pi_arc[(size_t)(j) * (size_t)L + (size_t)(i)]
    if (phase == PH_SW_INIT) {
        if (tid == 0) {
            S->rTr = d[0]; S->dot_r = d[0]; S->dot_zero = d[0];                      // (:344, 359-360)
            S->alpha = 1.0; S->beta = 0.0; S->omega = 0.0; S->rTr_old = 0.0; S->paused = 0;
            H->a_arc[0] = 1.0; H->b_arc[0] = 0.0;                                    // (:363-364)
            H->stop_count = 0; H->max_sigma = 0; H->switches = 0; H->r_scale = 1.0;
            if (!(0 < nsig && 0 < S->max_iter)) S->done = 1;
        }
        for (int j = tid; j < nsig; j += kBlock) {                                   // (:347-355)
            H->alpha[j] = 1.0; H->beta[j] = 0.0; H->eta[j] = 0.0; H->zeta[j] = 1.0; H->omega[j] = 0.0;
            PI_(j, 0) = 1.0; PI_(j, 1) = 1.0;
            H->cp[j] = 0.0; H->cx[j] = 0.0; H->c1[j] = 0.0; H->c2[j] = 0.0;
            H->stop[j] = 0; H->skip[j] = 0;
        }
        return;
    }
    if (phase == PH_SW_ALPHA) {
        if (tid == 0) { S->alpha = S->rTr / d[0]; H->a_arc[kk] = S->alpha; }        // (:392)
        return;
    }
    if (phase == PH_SW_OMEGA) {
        if (tid == 0) { S->omega = d[1] / d[0]; H->w_arc[kk] = S->omega; }          // (q,q)/(q,y)  (:412)
        return;
    }
    if (phase == PH_SW_END) {
        if (tid == 0) {
            S->dot_r = d[0];                                                         // (:416)
            S->rTr_old = S->rTr;
            S->rTr = d[1];                                                           // (:418)
            S->beta = (S->alpha / S->omega) * (S->rTr / S->rTr_old);                 // (:422)
            H->b_arc[kk] = S->beta;
        }
        __syncthreads();
        const int seed = H->seed;
        const double a = H->a_arc[kk], w = H->w_arc[kk], b = H->b_arc[kk];
        const double ratio = H->b_arc[kk - 1] / H->a_arc[kk - 1];
        const double sgs = H->sigma[seed];
        for (int j = tid; j < nsig; j += kBlock) {                                   // (:431-446)
            if (j == seed || H->stop[j]) { H->skip[j] = 1; continue; }
            const double dsg = sgs - H->sigma[j];
            const double po = PI_(j, kk - 1);
            const double e = ratio * a * H->eta[j] - dsg * a * po;                   // (:433)
            H->eta[j] = e;
            const double pn = e + po;                                                // (:435)
            PI_(j, kk) = pn;
            const double aj = (po / pn) * a;                                         // (:436)
            H->alpha[j] = aj;
            const double wj = w / (1.0 - w * dsg);                                   // (:437)
            H->omega[j] = wj;
            const double z = H->zeta[j];
            H->cx[j] = wj / (pn * z);                                                // (:438)
            H->c1[j] = wj / (aj * z * pn);                                           // (:440)
            H->c2[j] = -wj / (aj * z * po);                                          // (:441)
            const double zn = (1.0 - w * dsg) * z;                                   // (:442)
            H->zeta[j] = zn;
            H->beta[j] = (po / pn) * (po / pn) * b;                                  // (:443)
            H->cp[j] = 1.0 / (pn * zn);                                              // (:445)
            H->skip[j] = 0;
        }
        return;
    }
    if (phase != PH_SW_STOP) return;

    // ---- stop flags, largest |1/(zeta pi)| among the shifts still running                    (:451-475)
    __shared__ double s_val[kBlock];
    __shared__ int s_idx[kBlock], s_cnt[kBlock];
    __shared__ int s_switch;
    const int seed = H->seed;
    double best = 1.0;              // max_zeta_pi starts at 1.0: only larger values move max_sigma
    int best_j = 0x7fffffff, newly = 0;
    for (int j = tid; j < nsig; j += kBlock) {
        if (H->stop[j]) continue;
        const double av = j == seed ? 1.0 : fabs(1.0 / (H->zeta[j] * PI_(j, kk)));
        if (av * av * S->dot_r <= S->tol2 * S->dot_zero) { H->stop[j] = 1; ++newly; }
        else if (av > best) { best = av; best_j = j; }
    }
    s_val[tid] = best; s_idx[tid] = best_j; s_cnt[tid] = newly;
    __syncthreads();
    for (int wdt = kBlock / 2; wdt > 0; wdt >>= 1) {
        if (tid < wdt) {
            s_cnt[tid] += s_cnt[tid + wdt];
            const double ov = s_val[tid + wdt];
            const int oj = s_idx[tid + wdt];
            if (ov > s_val[tid] || (ov == s_val[tid] && oj < s_idx[tid])) { s_val[tid] = ov; s_idx[tid] = oj; }   // first j wins ties
        }
        __syncthreads();
    }
    if (tid == 0) {
        H->stop_count += s_cnt[0];
        if (s_val[0] > 1.0) H->max_sigma = s_idx[0];     // otherwise it keeps its previous value, as in the reference
        S->k += 1;                                       // (:536)
        const int k = S->k;
        if (H->unsolved_arc && k < H->arc_len) H->unsolved_arc[k] = nsig - H->stop_count;
        if (S->tr_dotr && k <= S->max_iter) {
            S->tr_alpha[k - 1] = S->alpha; S->tr_omega[k - 1] = S->omega; S->tr_beta[k - 1] = S->beta; S->tr_dotr[k - 1] = S->dot_r;
        }
        const bool more = H->stop_count < nsig && k < S->max_iter;                   // (:374 / :100)
        if (!more) S->done = 1;
        if (!(isfinite(S->alpha) && isfinite(S->beta) && isfinite(S->omega) && isfinite(S->dot_r)) && !S->breakdown_k)
            S->breakdown_k = k;
        s_switch = (H->mode == SH_SWITCH && H->stop[seed] && H->stop_count < nsig) ? 1 : 0;   // (:490)
        if (s_switch) {
            // the host has to rescale r and re-point the seed vectors: stop the device here
            S->done = 1;
            S->paused = more ? 1 : 2;
            const int ms = H->max_sigma;
            H->r_scale = 1.0 / (H->zeta[ms] * PI_(ms, kk));                          // (:499)
        }
    }
    __syncthreads();
    if (!s_switch) return;

    // ---- seed switching: rewrite the history for the new seed ms                             (:494-521)
    const int ms = H->max_sigma;
    const double dss = H->sigma[seed] - H->sigma[ms];
    for (int i = 1 + tid; i <= kk; i += kBlock) {                                    // (:494-498) entries are independent
        const double qn = PI_(ms, i - 1) / PI_(ms, i);
        H->a_arc[i] = qn * H->a_arc[i];
        H->b_arc[i] = qn * qn * H->b_arc[i];
        H->w_arc[i] = H->w_arc[i] / (1.0 - H->w_arc[i] * dss);
    }
    for (int j = tid; j < nsig; j += kBlock) { H->eta[j] = 0.0; H->zeta[j] = 1.0; }  // (:501-505)
    __syncthreads();
    const double sgm = H->sigma[ms];
    for (int j = tid; j < nsig; j += kBlock) {                                       // (:509-518) one shift per thread
        if (H->stop[j] || j == ms) continue;
        double e = 0.0, z = 1.0;
        const double dsg = sgm - H->sigma[j];
        for (int i = 1; i <= kk; ++i) {
            const double ai = H->a_arc[i];
            e = (H->b_arc[i - 1] / H->a_arc[i - 1]) * ai * e - dsg * ai * PI_(j, i - 1);
            PI_(j, i) = e + PI_(j, i - 1);
            z = (1.0 - H->w_arc[i] * dsg) * z;
        }
        H->eta[j] = e; H->zeta[j] = z;
    }
    __syncthreads();
    if (tid == 0) { H->seed = ms; H->switches += 1; }                                // (:525)
#undef PI_
}

// whole-workgroup entry: scalar phases run on thread 0, shifted phases on all threads.
// HEAVY = false leaves out the seed-switching phases (and, in reduce_publish, the peer-to-peer
// in-kernel collect): inlined into every dot-producing kernel they cost the hot single-GPU kernels
// 26 VGPRs and 8 KB of LDS (occupancy 8 -> 5 waves per SIMD, +2-3 % per iteration on Transport), so
// the launch wrappers pick the HEAVY instantiation only for launches that need it (heavy_needed).
template <bool HEAVY>
__device__ __forceinline__ void apply_phase_block(Scal *S, int phase)
{
    if (HEAVY && phase >= PH_SW_INIT) apply_phase_switching(S, phase);
    else if (phase >= PH_SH_INIT && phase < PH_SW_INIT) apply_phase_shifted<kBlock>(S, phase);
    else if (phase < PH_SH_INIT && threadIdx.x == 0) apply_phase(S, phase);
}

static inline bool heavy_needed(const Reduce &red)
{
    return red.apply_now && (red.p2p.seq != 0 || red.phase >= PH_SW_INIT);
}

// Collect group pr.seq from this rank's mailbox (all P sources), leave the sums in Scal::red and
// apply the phase. Returns false (and raises comm_error/done) on a time-out.
__device__ __forceinline__ bool p2p_collect(Scal *S, int n, const P2pRed &pr, unsigned long long timeout_ticks, double *vals,
                                            int *s_fail)
{
    if (threadIdx.x == 0) *s_fail = 0;
    __syncthreads();
    const llword *mine = pr.mail[pr.rank];
    for (int t = threadIdx.x; t < n * pr.nranks; t += kBlock) {
        const int p = t / n, d = t % n;
        double v;
        if (!ll_wait(mine + mail_index(pr.seq, pr.nranks, p, d), pr.seq, timeout_ticks, &v)) *s_fail = 1;
        vals[p * kRedSlots + d] = v;
    }
    __syncthreads();
    if (*s_fail) {
        if (threadIdx.x == 0) { S->comm_error = 1; S->done = 1; }
        return false;
    }
    if ((int)threadIdx.x < n) S->red[threadIdx.x] = rank_tree_sum(vals + threadIdx.x, pr.nranks);
    __syncthreads();
    return true;
}

// Self-test of the transport: `rounds` all-reduces of five values that depend on (rank, round,
// slot), each checked against the sum recomputed locally. One workgroup; status[0] counts wrong
// sums, status[1] time-outs.
__device__ __forceinline__ double selftest_value(int rank, unsigned seq, int d)
{
    return (double)(rank * 131 + d * 17 + 1) * 1.000000119 + (double)seq * 0.333333333333;
}

// ------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------
// Sum ND values over the workgroup; every thread receives the totals in v[].
template <int ND>
__device__ __forceinline__ void block_sum(double (&v)[ND], double *sm /* [4*ND + ND] */)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < ND; ++d) v[d] = wave_sum(v[d]);
    __syncthreads();   // sm may still be read from a previous use
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < ND; ++d) sm[w * ND + d] = v[d];
    }
    __syncthreads();
    if (threadIdx.x < ND) {
        double s = sm[threadIdx.x];
#pragma unroll
        for (int ww = 1; ww < kBlock / 64; ++ww) s += sm[ww * ND + threadIdx.x];
        sm[4 * ND + threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < ND; ++d) v[d] = sm[4 * ND + d];
}

// Publish this workgroup's ND partial sums into slot `slot`. Arrival is counted in kShards sharded
// counters (a single word would serialise ~2000 simultaneous arrivals at ~12 ns each); the last
// arriver of a shard sums that shard's partials, the last shard to finish sums the shard totals,
// stores them in Scal::red and (single rank) applies the scalar recurrence. Membership and order
// of every sum are fixed by the slot numbers, so the result is deterministic.
// Cross-workgroup visibility follows the gfx950 rules for 8-byte agent-scope atomics on both
// sides: write-through (sc1) stores drained with an explicit vmcnt(0) before the ticket, sc1
// loads (L1 bypass) in the summing workgroup, one agent acquire fence before them.
__device__ __forceinline__ unsigned shard_population(unsigned expected, unsigned shard)
{
    return shard < expected ? (expected - shard + kShards - 1) / kShards : 0u;
}

template <int ND, bool HEAVY>
__device__ __forceinline__ void reduce_publish(double (&acc)[ND], Scal *S, const Reduce &red, unsigned slot, double *sm,
                                               unsigned order = 0xFFFFFFFFu)
{
    // slot: where this workgroup's partial goes (fixes the association of the group's sum); order: its position in LAUNCH
    // order, which decides who stays behind to finish (tail finish: the workgroups launched last). They differ when a launch
    // visits its row groups in another order than its workgroups are numbered (SpmvArgs::reverse / xcd_map): the partial of a
    // row group lands in the same slot whatever the order, so the sums keep their bits.
    if (order == 0xFFFFFFFFu) order = slot;
    __shared__ unsigned s_last;
    const unsigned shard = slot % kShards;
    block_sum<ND>(acc, sm);
    if (!HEAVY && red.tail_seq) {
        // ---- tail finish: LL-tagged partials, producers leave at once (struct Reduce)
        const unsigned seq = red.tail_seq, nsh = red.expected < (unsigned)kShards ? red.expected : (unsigned)kShards;
        if (threadIdx.x < ND) ll_store_agent(red.tail_tab + (size_t)slot * kTailStride + 2 * threadIdx.x, acc[threadIdx.x], seq);
        if (order + nsh < red.expected) return;
        const unsigned sh = red.expected - 1u - order;                       // this workgroup's shard: 0 = the very last workgroup
        const unsigned long long patience = 400000000ull;                   // 4 s: a lost workgroup must not hang the GPU
        double tot[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) tot[d] = 0.0;
        bool lost = false;
        for (unsigned i = sh + threadIdx.x * nsh; i < red.expected && !lost; i += kBlock * nsh) {     // slots sh, sh + nsh, ... in order
            const llword *row = red.tail_tab + (size_t)i * kTailStride;
            const unsigned long long t0 = wall_clock64();
            for (unsigned spin = 0;; ++spin) {
                double v[ND];
                bool all = true;
#pragma unroll
                for (int d = 0; d < ND; ++d) all = ll_peek_agent(row + 2 * d, seq, &v[d]) && all;
                if (all) {
#pragma unroll
                    for (int d = 0; d < ND; ++d) tot[d] += v[d];
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if ((spin & 63u) == 63u && wall_clock64() - t0 > patience) { lost = true; break; }
            }
        }
        // a producer that never delivered: the error is raised by the very thread that gave up (any thread of any shard
        // workgroup, not only thread 0 of the last one), and its shard total is poisoned -- a NaN cannot pass for a sum
        if (lost) { S->comm_error = 1; S->done = 1; tot[0] = __builtin_nan(""); }
        block_sum<ND>(tot, sm);
        if (threadIdx.x < ND) ll_store_agent(red.tail_shard + ((size_t)sh * kRedSlots + threadIdx.x) * 2, tot[threadIdx.x], seq);
        if (sh != 0) return;
#pragma unroll
        for (int d = 0; d < ND; ++d) tot[d] = 0.0;
        if (threadIdx.x < nsh) {
            const unsigned long long t0 = wall_clock64();
            for (unsigned spin = 0;; ++spin) {
                double v[ND];
                bool all = true;
#pragma unroll
                for (int d = 0; d < ND; ++d) all = ll_peek_agent(red.tail_shard + ((size_t)threadIdx.x * kRedSlots + d) * 2, seq, &v[d]) && all;
                if (all) {
#pragma unroll
                    for (int d = 0; d < ND; ++d) tot[d] = v[d];
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                if ((spin & 63u) == 63u && wall_clock64() - t0 > patience) { lost = true; break; }
            }
        }
        if (lost) { S->comm_error = 1; S->done = 1; tot[0] = __builtin_nan(""); }      // any thread's wait, not only thread 0's
        block_sum<ND>(tot, sm);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int d = 0; d < ND; ++d) S->red[red.red_off + d] = tot[d];
        }
        if (red.apply_now) {
            __syncthreads();
            apply_phase_block<HEAVY>(S, red.phase);
        }
        return;
    }
    if (threadIdx.x < ND)
        __hip_atomic_store(&red.partial[(size_t)slot * kPartialStride + threadIdx.x], acc[threadIdx.x],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(&red.counter[shard * kCounterStride], 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == shard_population(red.expected, shard) - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;

    // ---- last arriver of this shard: sum the shard's partials in slot order
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    const unsigned members = shard_population(red.expected, shard);
    double tot[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) tot[d] = 0.0;
    for (unsigned i = threadIdx.x; i < members; i += kBlock) {
        const size_t sl = (size_t)shard + (size_t)i * kShards;
#pragma unroll
        for (int d = 0; d < ND; ++d)
            tot[d] += __hip_atomic_load(&red.partial[sl * kPartialStride + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    block_sum<ND>(tot, sm);
    if (threadIdx.x < ND)
        __hip_atomic_store(&red.shard_tot[shard * kPartialStride + threadIdx.x], tot[threadIdx.x], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    const unsigned active = red.expected < (unsigned)kShards ? red.expected : (unsigned)kShards;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&red.counter[shard * kCounterStride], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(&red.counter[kShards * kCounterStride], 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == active - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;

    // ---- last shard: sum the shard totals in shard order, finish the group
#pragma unroll
    for (int d = 0; d < ND; ++d)
        tot[d] = threadIdx.x < active ? __hip_atomic_load(&red.shard_tot[threadIdx.x * kPartialStride + d], __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT)
                                      : 0.0;
    block_sum<ND>(tot, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < ND; ++d) S->red[red.red_off + d] = tot[d];
        __hip_atomic_store(&red.counter[kShards * kCounterStride], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (red.p2p.seq) {
        // peer-to-peer all-reduce: hand the finished local sums to every rank's mailbox (block_sum
        // left them in sm[4*ND + d])
        for (int t = threadIdx.x; t < ND * red.p2p.nranks; t += kBlock) {
            const int d = t % ND, p = t / ND;
            if ((red.p2p.mask >> d) & 1u)
                ll_store(red.p2p.mail[p] + mail_index(red.p2p.seq, red.p2p.nranks, red.p2p.rank, red.red_off + d),
                         sm[4 * ND + d], red.p2p.seq);
        }
    }
    if (red.apply_now) {
        if (HEAVY && red.p2p.seq) {
            // peer-to-peer, group not deferred: this workgroup is the last of the launch anyway, so it
            // waits for the other ranks' sums and applies the phase right here instead of leaving
            // that to a separate one-workgroup kernel (3-4 us per dot group on a small rank)
            __shared__ double p2p_vals[kRedSlots * kMaxRanksP2p];
            __shared__ int p2p_fail;
            if (!p2p_collect(S, red.p2p.n_collect, red.p2p, red.p2p.timeout_ticks, p2p_vals, &p2p_fail)) return;
        } else {
            __syncthreads();        // Scal::red written by thread 0 above
        }
        apply_phase_block<HEAVY>(S, red.phase);
    }
}


// ------------------------------------------------------------------------------------------
// consumer-side finish of a dot group (struct Finish, bicg_device.h)
// ------------------------------------------------------------------------------------------
// producer epilogue: one partial per WAVEFRONT -- shuffle, one plain store per sum, done. No
// barrier, no atomic: the kernel boundary in front of the consuming kernel makes it visible.
template <int ND>
__device__ __forceinline__ void wave_publish(double (&acc)[ND], double *partial, unsigned wg_slot)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = wave_sum(acc[d]);
    if (lane == 0) {
        double *row = partial + ((size_t)wg_slot * (kBlock / 64) + wave) * kPartialStride;
#pragma unroll
        for (int d = 0; d < ND; ++d) row[d] = acc[d];
    }
}

struct FinishLds {
    double vals[kShards * kMaxDots];          // shard totals
    double pv[kRedSlots * kMaxRanksP2p];      // peer-to-peer: every rank's sums
    double sums[kRedSlots];
    double bs[5 * kMaxDots];                  // block_sum scratch
    unsigned missing;
    int fail;
};

// whole workgroup: add up shard `shard` of the group's partials (slot order) and publish the total
__device__ __forceinline__ void finish_sum_shard(const Finish &f, unsigned shard, FinishLds &L)
{
    const unsigned members = shard < f.nparts ? (f.nparts - shard + kShards - 1) / kShards : 0u;
    double tot[kMaxDots];
#pragma unroll
    for (int d = 0; d < kMaxDots; ++d) tot[d] = 0.0;
    for (unsigned i = threadIdx.x; i < members; i += kBlock) {
        const double *row = f.partial + ((size_t)shard + (size_t)i * kShards) * kPartialStride;
#pragma unroll
        for (int d = 0; d < kMaxDots; ++d)
            if (d < f.n) tot[d] += row[d];
    }
    block_sum<kMaxDots>(tot, L.bs);
    if ((int)threadIdx.x < f.n)
        ll_store_agent(f.shard + ((size_t)shard * kRedSlots + threadIdx.x) * 2, tot[threadIdx.x], f.seq);
}

// whole workgroup: L.sums[0..n) <- the group's sums over this GPU's partials and (peer-to-peer) over
// all ranks. Nobody is waited for longer than spin_ticks inside the GPU: a shard total that has not
// shown up by then is computed here (same partials, same order, same bits), so the result does
// not depend on the order in which the hardware dispatches workgroups. False: a PEER timed out.
// (got0, v0): outcome of a first look at this thread's shard total that the caller has already taken.
__device__ __forceinline__ bool finish_totals(const Finish &f, int roles, FinishLds &L, bool block0, bool got0, double v0)
{
    const unsigned tid = threadIdx.x;
    const bool exchange = f.p2p.seq != 0 && !(roles & FIN_LOCAL);
    if (!exchange || (roles & FIN_PUSH)) {
        for (bool first = true;; first = false) {
            if (tid == 0) L.missing = 0u;
            __syncthreads();
            if ((int)tid < kShards * f.n) {
                const int sh = (int)tid / f.n, d = (int)tid % f.n;
                double v = v0;
                if ((first && got0) || ll_try_agent(f.shard + ((size_t)sh * kRedSlots + d) * 2, f.seq, f.spin_ticks, &v))
                    L.vals[sh * kMaxDots + d] = v;
                else atomicOr(&L.missing, 1u << sh);
            }
            __syncthreads();
            const unsigned m = L.missing;
            if (!m) break;
            for (unsigned sh = 0; sh < (unsigned)kShards; ++sh)
                if ((m >> sh) & 1u) finish_sum_shard(f, sh, L);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if ((int)tid < f.n) {
            double t = 0.0;
            for (int sh = 0; sh < kShards; ++sh) t += L.vals[sh * kMaxDots + tid];
            L.sums[tid] = t;
        }
        __syncthreads();
    }
    if (!exchange) return true;
    const int P = f.p2p.nranks;
    if ((roles & FIN_PUSH) && block0)
        for (int t = tid; t < f.n * P; t += kBlock) {
            const int p = t / f.n, d = t % f.n;
            ll_store(f.p2p.mail[p] + mail_index(f.p2p.seq, P, f.p2p.rank, f.red_off + d), L.sums[d], f.p2p.seq);
        }
    if (!(roles & (FIN_APPLY | FIN_BLOCK0))) return true;
    if (tid == 0) L.fail = 0;
    __syncthreads();
    const llword *mine = f.p2p.mail[f.p2p.rank];
    for (int t = tid; t < f.n * P; t += kBlock) {
        const int p = t / f.n, d = t % f.n;
        double v;
        if (p == f.p2p.rank && (roles & FIN_PUSH)) v = L.sums[d];
        else if (!ll_wait(mine + mail_index(f.p2p.seq, P, p, f.red_off + d), f.p2p.seq, f.p2p.timeout_ticks, &v)) L.fail = 1;
        L.pv[p * kRedSlots + d] = v;
    }
    __syncthreads();
    if (L.fail) return false;
    if ((int)tid < f.n) L.sums[tid] = rank_tree_sum(L.pv + tid, P);
    __syncthreads();
    return true;
}

// Prologue of a kernel that carries a Finish. bid / nblocks: this workgroup's index among the
// workgroups that take part (SpMV launches exclude their leading halo-push workgroups).
// Returns the scalar block the kernel has to read: S itself, or -- FIN_APPLY -- the private copy
// *priv (LDS) on which the recurrence has been applied; workgroup 0 has then written it to
// f.Snext as well, also when the solve is already `done` (the host switches blocks regardless).
// roles: what THIS call does of roles (a launch may sum the shards at its start and apply at its epilogue).
__device__ __forceinline__ const Scal *finish_group(Scal *S, const Finish &f, int roles, unsigned bid, unsigned nblocks, FinishLds &L,
                                                    Scal *priv)
{
    const unsigned tid = threadIdx.x;
    const bool all = (roles & FIN_APPLY) != 0, block0 = bid == 0;
    // everything this prologue may need from memory is requested before the first of it is looked at:
    // `done`, the alarm, a copy of the scalar block (one wavefront) and this thread's shard total --
    // one round trip instead of four dependent ones underneath the kernel's own loads
    const int done = S->done;
    const int alarm = f.alarm ? *f.alarm : 0;
    double v0 = 0.0;
    bool got0 = false;
    const bool local = !(f.p2p.seq != 0 && !(roles & FIN_LOCAL)) || (roles & FIN_PUSH);
    if (all && local && !(roles & FIN_SHARDS) && (int)tid < kShards * f.n)
        got0 = ll_peek_agent(f.shard + ((size_t)((int)tid / f.n) * kRedSlots + (int)tid % f.n) * 2, f.seq, &v0);
    if (all && tid == 64) *priv = *S;
    if (done) {          // converged: producers wrote nothing, nothing may change any more
        if (all && block0 && tid == 0) *f.Snext = *S;
        return S;
    }
    if (block0 && f.shard_clear)
        for (unsigned t = tid; t < (unsigned)(kShardLL * kRedSlots * 2); t += kBlock) f.shard_clear[t] = 0ull;
    if (alarm) {           // a peer was lost earlier: nothing will ever arrive
        if (all) {
            __syncthreads();
            if (tid == 0) { priv->done = 1; priv->comm_error = 1; if (block0) *f.Snext = *priv; }
            __syncthreads();
            // the workgroups that take the applied scalars from this one must not sit through their own peer time-out
            // (one per occupancy wave of row workgroups): the row is published all the same, with done = 1
            if (block0 && tid < 4)
                ll_store_agent(f.shard + (size_t)kShards * kRedSlots * 2 + 2 * tid, tid == 3 ? 1.0 : 0.0, f.seq);
            return priv;
        }
        return S;
    }
    if (roles & FIN_SHARDS)
        for (unsigned sh = bid; sh < (unsigned)kShards; sh += nblocks) finish_sum_shard(f, sh, L);
    const bool consume = all || (block0 && (roles & (FIN_BLOCK0 | FIN_PUSH)));
    if (!consume) return S;
    llword *const row = f.shard + (size_t)kShards * kRedSlots * 2;      // the applied scalars as workgroup 0 publishes them
    if (all && !block0 && f.p2p.seq != 0) {
        // Sums that cross GPUs are collected by ONE workgroup (the first of the launch: it waits for peers, never for
        // this launch); everybody else takes the applied scalars from it. Hundreds of workgroups polling the mailbox
        // in uncached memory would crowd out the very stores they are waiting for.
        if (tid == 0) L.fail = 0;
        __syncthreads();
        if (tid < 4) {
            double v;
            if (ll_wait_agent(row + 2 * tid, f.seq, f.p2p.timeout_ticks, &v)) L.sums[tid] = v;
            else L.fail = 1;
        }
        __syncthreads();
        if (tid == 0) {
            if (L.fail) { if (f.alarm) *f.alarm = 1; priv->done = 1; priv->comm_error = 1; }
            else { priv->alpha = L.sums[0]; priv->beta = L.sums[1]; priv->omega = L.sums[2]; priv->done = L.sums[3] != 0.0 ? 1 : 0; }
        }
        __syncthreads();
        return priv;
    }
    const bool ok = finish_totals(f, roles, L, block0, got0, v0);
    if (!(roles & (FIN_APPLY | FIN_BLOCK0))) return S;
    if (tid == 0) {
        if (!ok && f.alarm) *f.alarm = 1;
        // all: the recurrence runs on the private copy in LDS (other workgroups of this launch may still
        // be reading S); otherwise (stand-alone finisher) workgroup 0 alone works in place
        Scal *T = all ? priv : S;
        if (ok) {
            for (int d = 0; d < f.n; ++d) T->red[f.red_off + d] = L.sums[d];
            if (f.phase != PH_NONE) apply_phase(T, f.phase, block0);
        } else {
            T->done = 1; T->comm_error = 1;
        }
        if (all && block0) *f.Snext = *priv;
    }
    if (!all) return S;
    __syncthreads();
    if (block0 && tid < 4) {
        const double v = tid == 0 ? priv->alpha : tid == 1 ? priv->beta : tid == 2 ? priv->omega : (double)priv->done;
        ll_store_agent(row + 2 * tid, v, f.seq);
    }
    return priv;
}


// which reduction epilogue / prologue a launch needs (RedMode)
static inline int red_mode(const Reduce &red, const Finish &fin, bool has_dots)
{
    if (fin.seq || (has_dots && red.wave)) return RED_WAVE;
    return has_dots && heavy_needed(red) ? RED_TICKET_HEAVY : RED_TICKET;
}

}  // namespace bicg
