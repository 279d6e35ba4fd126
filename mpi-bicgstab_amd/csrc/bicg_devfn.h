// bicg_devfn.h -- device functions shared by the kernel translation units (bicg_kernels.hip, bicg_persist.hip):
// the scalar recurrences of the four solvers, LL words (payload + sequence tag in one 8-byte store), DPP wavefront sums.
// gfx950 only; compiled with -ffp-contract=off like everything else.
#pragma once

#include "bicg_device.h"

namespace bicg {

// ------------------------------------------------------------------------------------------
// scalar recurrences (one thread)
// ------------------------------------------------------------------------------------------
// publish = false: the caller works on a private copy of the scalar block (consumer-side finish);
// the per-iteration trace in global memory is written by the one workgroup that publishes
__device__ __forceinline__ void finish_iteration(Scal *S, double alpha_used, bool publish)
{
    S->k += 1;
    const int k = S->k;
    if (publish && S->tr_dotr && k <= S->max_iter) {
        S->tr_alpha[k - 1] = alpha_used;
        S->tr_omega[k - 1] = S->omega;
        S->tr_beta[k - 1]  = S->beta;
        S->tr_dotr[k - 1]  = S->dot_r;
    }
    // reference loop condition, src/solver.c:86 / 216 / 351
    if (!(S->dot_r > S->tol2 * S->dot_zero && k < S->max_iter)) S->done = 1;
    // Breakdown guard: the reference keeps iterating on NaNs until
    // MAX_ITER (its loop condition is false for NaN only by accident of the comparison); here a
    // non-finite recurrence scalar is recorded -- the iteration count and vectors are left as the
    // reference would leave them at this k.
    if (!(isfinite(S->alpha) && isfinite(S->beta) && isfinite(S->omega) && isfinite(S->dot_r))) {
        if (!S->breakdown_k) S->breakdown_k = k;
    }
}

__device__ __forceinline__ void apply_phase(Scal *S, int phase, bool publish = true)
{
    const double *d = S->red;
    switch (phase) {
    case PH_INIT:
        S->rTr = d[0]; S->dot_r = d[0]; S->dot_zero = d[0];
        S->alpha = 0.0; S->beta = 0.0; S->omega = 0.0; S->rTr_old = 0.0;
        if (!(S->dot_r > S->tol2 * S->dot_zero && 0 < S->max_iter)) S->done = 1;
        break;
    case PH_INIT_ALPHA:
        S->alpha = S->rTr / d[0]; S->beta = 0.0; S->omega = 0.0;
        break;
    case PH_PLAIN_ALPHA:
        S->alpha = S->rTr / d[0];
        break;
    case PH_OMEGA:
        S->omega = d[0] / d[1];
        break;
    case PH_PLAIN_END: {
        S->dot_r = d[0];
        S->rTr_old = S->rTr;
        S->rTr = d[1];
        S->beta = (S->alpha / S->omega) * (S->rTr / S->rTr_old);
        finish_iteration(S, S->alpha, publish);
        break;
    }
    case PH_RECUR_END: {
        const double alpha_used = S->alpha;
        S->dot_r = d[0];
        S->rTr_old = S->rTr;
        S->rTr = d[1];
        const double rTw = d[2], rTs = d[3], rTz = d[4];
        S->beta = (S->alpha / S->omega) * (S->rTr / S->rTr_old);
        S->alpha = S->rTr / (rTw + S->beta * (rTs - S->omega * rTz));
        finish_iteration(S, alpha_used, publish);
        break;
    }
    default: break;
    }
}


// ------------------------------------------------------------------------------------------
// peer-to-peer transport: LL words (see bicg_device.h)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void ll_store(llword *dst, double v, unsigned seq)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned long long tag = (unsigned long long)seq << 32;
    __hip_atomic_store(dst, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Spin until both words carry `seq`; false after timeout_ticks of the 100 MHz wall clock (a peer
// that died or diverged must not hang the GPU).
__device__ __forceinline__ bool ll_wait(const llword *src, unsigned seq, unsigned long long timeout_ticks, double *out)
{
    const unsigned long long t0 = wall_clock64();
    for (unsigned spin = 0;; ++spin) {
        const unsigned long long w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(w0 >> 32) == seq && (unsigned)(w1 >> 32) == seq) {
            *out = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
            return true;
        }
        if ((spin & 63u) == 63u) {
            if (wall_clock64() - t0 > timeout_ticks) { *out = 0.0; return false; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
}

// Sum the P contributions of every value the way a recursive-doubling all-reduce associates them
// ((v0+v1)+(v2+v3))+... : fixed order => every rank computes bit-identical sums.
__device__ __forceinline__ double rank_tree_sum(double *v /* [nranks], stride kRedSlots */, int nranks)
{
    for (int stride = 1; stride < nranks; stride <<= 1)
        for (int i = 0; i + stride < nranks; i += 2 * stride) v[i * kRedSlots] += v[(i + stride) * kRedSlots];
    return v[0];
}


// Sum over the 64 lanes of a wavefront with DPP lane permutations (quad swaps, row mirrors, row
// broadcasts): plain VALU moves, where __shfl_down goes through the LDS crossbar (ds_bpermute, two per
// double and step: ~0.4 us per sum at the end of every dot-producing workgroup). Fixed association;
// every lane receives the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ROW_MASK == 0xF) {       // every lane has a source: no previous value to keep, no register to clear first
        lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);     // rows outside ROW_MASK receive 0: v + 0.0
        hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    }
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]: pairs
    v = dpp_add<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]: quads
    v = dpp_add<0x141, 0xF>(v);     // row_half_mirror: 8 lanes
    v = dpp_add<0x140, 0xF>(v);     // row_mirror: rows of 16
    v = dpp_add<0x142, 0xA>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);     // row_bcast:31 into rows 2 and 3: lane 63 holds the wavefront's sum
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}


// LL words inside one GPU (agent scope): shard totals travel from the summing workgroup to every
// workgroup of the same launch
__device__ __forceinline__ void ll_store_agent(llword *dst, double v, unsigned seq)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned long long tag = (unsigned long long)seq << 32;
    __hip_atomic_store(dst, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one look at a word pair: true (and the value) when both carry `seq`
__device__ __forceinline__ bool ll_peek_agent(const llword *src, unsigned seq, double *out)
{
    const unsigned long long w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *out = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
    return (unsigned)(w0 >> 32) == seq && (unsigned)(w1 >> 32) == seq;
}
__device__ __forceinline__ bool ll_try_agent(const llword *src, unsigned seq, unsigned long long ticks, double *out)
{
    const unsigned long long t0 = wall_clock64();
    for (unsigned spin = 0;; ++spin) {
        if (ll_peek_agent(src, seq, out)) return true;
        if ((spin & 15u) == 15u) {
            if (wall_clock64() - t0 > ticks) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// long wait with back-off (a value that another GPU's sums have to arrive for first): polls get rarer the longer it takes
__device__ __forceinline__ bool ll_wait_agent(const llword *src, unsigned seq, unsigned long long ticks, double *out)
{
    const unsigned long long t0 = wall_clock64();
    for (unsigned spin = 0;; ++spin) {
        if (ll_peek_agent(src, seq, out)) return true;
        if (spin < 32u) { __builtin_amdgcn_s_sleep(1); continue; }
        if (wall_clock64() - t0 > ticks) return false;
        if (spin < 256u) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(64);
    }
}


// ---- shifted solvers: per-shift scalar recurrences, one thread per shift (whole workgroup calls)
// lop / pipe: the per-shift block of reference src/shifted_solver.c:264-301 (= :803-839)
__device__ __forceinline__ void shift_beta_cp(ShiftDev *H, int j, double beta_seed)
{
    const double po = H->pi_old[j], pn = H->pi_new[j];
    H->beta[j] = (po / pn) * (po / pn) * beta_seed;                  // (:264)
    H->cp[j] = 1.0 / (pn * H->zeta[j]);                              // (:266)
    H->pi_old[j] = pn;                                               // (:268)
}
__device__ __forceinline__ void shift_eta_alpha(ShiftDev *H, int j, double a_seed, double sg_seed)
{
    const double pn = H->pi_old[j];                                  // already copied
    const double e = (H->beta_old / H->alpha_old) * a_seed * H->eta[j] - (sg_seed - H->sigma[j]) * a_seed * pn;   // (:283)
    H->eta[j] = e;
    const double pnew = e + pn;                                      // (:285)
    H->pi_new[j] = pnew;
    H->alpha[j] = (pn / pnew) * a_seed;                              // (:286)
}
__device__ __forceinline__ void shift_omega_coeffs(ShiftDev *H, int j, double w_seed, double sg_seed)
{
    const double dsg = sg_seed - H->sigma[j];
    const double wj = w_seed / (1.0 - w_seed * dsg);                 // (:295)
    H->omega[j] = wj;
    const double pn = H->pi_new[j], po = H->pi_old[j], z = H->zeta[j], aj = H->alpha[j];
    H->cx[j] = wj / (pn * z);                                        // (:296)
    H->c1[j] = wj / (aj * z * pn);                                   // (:298)
    H->c2[j] = -wj / (aj * z * po);                                  // (:299)
    H->zeta[j] = (1.0 - w_seed * dsg) * z;                           // (:300)
}

// MAXT >= blockDim.x (a power of two); kernels of bicg_kernels.hip call it with 256 threads, the persistent kernels' helper
// workgroup with up to 1024. The only reduction is a maximum: the result does not depend on the number of threads.
template <int MAXT>
__device__ __forceinline__ void apply_phase_shifted(Scal *S, int phase, double *smax_buf = nullptr /* [MAXT] LDS, or the static one */)
{
    const int nthr = (int)blockDim.x;
    ShiftDev *H = S->sh;
    const double *d = S->red;
    const int nsig = H->nsig, seed = H->seed, mode = H->mode;
    __shared__ double s_max_static[MAXT <= 256 ? MAXT : 1];
    double *const s_max = (MAXT <= 256 && !smax_buf) ? s_max_static : smax_buf;
    if (threadIdx.x == 0) {
        switch (phase) {
        case PH_SH_INIT:
            S->rTr = d[0]; S->dot_r = d[0]; S->dot_zero = d[0];
            S->alpha = 1.0; S->beta = 0.0; S->omega = 0.0; S->rTr_old = 0.0;
            H->max_zeta_pi = 1.0; H->alpha_old = 1.0; H->beta_old = 0.0;
            if (!(1.0 * 1.0 * S->dot_r > S->tol2 * S->dot_zero && 0 < S->max_iter)) S->done = 1;
            break;
        case PH_SH_ALPHA:
            H->alpha_old = S->alpha;                  // alpha_old <- alpha[seed]   (:270 / :96)
            H->beta_old = S->beta;                    // beta_old  <- beta[seed]    (:271 / :97)
            S->alpha = S->rTr / d[0];                 // alpha[seed] <- (r#,r)/(r#,s)  (:274 / :100)
            break;
        case PH_SH_OMEGA:
            S->omega = mode == SH_XI ? d[0] / d[1]    // (q,y)/(y,y)                (:115)
                                     : d[1] / d[0];   // (q,q)/(q,y)                (:291)
            break;
        case PH_SH_END:
            S->dot_r = d[0];
            S->rTr_old = S->rTr;
            S->rTr = d[1];
            S->beta = (S->alpha / S->omega) * (S->rTr / S->rTr_old);      // (:310 / :135)
            break;
        case PH_SHP_INIT_ALPHA:
            H->alpha_old = 1.0;                       // (:785)
            S->alpha = S->rTr / d[0];                 // (:786)
            break;
        case PH_SHP_OMEGA:
            H->beta_old = S->beta;                    // (:817)
            S->omega = d[0] / d[1];                   // (q,y)/(y,y)                (:828)
            break;
        case PH_SHP_END: {
            S->dot_r = d[0];
            S->rTr_old = S->rTr;
            S->rTr = d[1];
            S->beta = (S->alpha / S->omega) * (S->rTr / S->rTr_old);      // (:856)
            H->alpha_old = S->alpha;                                      // (:857)
            S->alpha = S->rTr / (d[2] + S->beta * (d[3] - S->omega * d[4]));   // (:858)
            break;
        }
        default: break;
        }
    }
    __syncthreads();
    const double a_seed = S->alpha, w_seed = S->omega, sg_seed = H->sigma[seed];
    double local_max = 1.0;
    for (int j = threadIdx.x; j < nsig; j += nthr) {
        if (phase == PH_SH_INIT) {
            H->beta[j] = 0.0; H->alpha[j] = 1.0; H->eta[j] = 0.0; H->pi_old[j] = 1.0; H->pi_new[j] = 1.0; H->zeta[j] = 1.0;
            H->cp[j] = 0.0; H->cx[j] = 0.0; H->c1[j] = 0.0; H->c2[j] = 0.0; H->omega[j] = 0.0;
            continue;
        }
        if (j == seed) {
            if (mode != SH_XI && (phase == PH_SH_ALPHA || phase == PH_SHP_OMEGA)) H->pi_old[j] = H->pi_new[j];   // my_dcopy copies every entry
            continue;
        }
        if (mode == SH_XI) {
            // xi_old = pi_old, xi_curr = pi_new, xi_new = eta, tau = zeta      (src/shifted_solver.c:90-142)
            const double xo = H->pi_old[j], xc = H->pi_new[j], tau = H->zeta[j], sg = H->sigma[j];
            if (phase == PH_SH_ALPHA) {
                H->beta[j] = (xc / xo) * (xc / xo) * H->beta_old;                        // (:91)
                H->cp[j] = tau * xc;                                                     // (:93)
            } else if (phase == PH_SH_OMEGA) {
                const double xn = (xc * xo * H->alpha_old) /
                                  (a_seed * H->beta_old * (xo - xc) + xo * H->alpha_old * (1.0 + a_seed * sg));   // (:108)
                H->eta[j] = xn;
                const double aj = (xn / xc) * a_seed;                                    // (:110)
                H->alpha[j] = aj;
                const double wj = w_seed / (1.0 + w_seed * sg);                          // (:119)
                H->omega[j] = wj;
                H->cx[j] = wj * tau * xn;                                                // (:120)
                H->c1[j] = wj * tau * xn / aj;                                           // (:122)
                H->c2[j] = -wj * tau * xc / aj;                                          // (:123)
            } else if (phase == PH_SH_END) {
                const double tn = tau / (1.0 + w_seed * sg);                             // (:130)
                H->zeta[j] = tn;
                double a = xc * tn;                                                      // (:138)
                if (a < 0.0) a = -a;
                if (a > local_max) local_max = a;
                H->pi_old[j] = xc;                                                       // (:141)
                H->pi_new[j] = H->eta[j];                                                // (:142)
            }
            continue;
        }
        if (phase == PH_SH_ALPHA) {
            shift_beta_cp(H, j, H->beta_old);        // beta[seed] of the previous iteration
            shift_eta_alpha(H, j, a_seed, sg_seed);
        } else if (phase == PH_SH_OMEGA) {
            shift_omega_coeffs(H, j, w_seed, sg_seed);
        } else if (phase == PH_SHP_OMEGA) {
            shift_beta_cp(H, j, H->beta_old);        // (:805-807), beta_old == beta[seed] here
            shift_eta_alpha(H, j, a_seed, sg_seed);  // (:820-823)
            shift_omega_coeffs(H, j, w_seed, sg_seed);   // (:833-838)
        } else if (phase == PH_SH_END || phase == PH_SHP_END) {
            double a = 1.0 / (H->zeta[j] * H->pi_new[j]);                // (:314 / :862)
            if (a < 0.0) a = -a;
            if (a > local_max) local_max = a;
        }
    }
    if (phase == PH_SH_END || phase == PH_SHP_END) {
        s_max[threadIdx.x] = local_max;
        for (int t = nthr + (int)threadIdx.x; t < MAXT; t += nthr) s_max[t] = 1.0;       // (local_max starts at 1.0 everywhere)
        __syncthreads();
        for (int w = MAXT / 2; w > 0; w >>= 1) {
            for (int t = (int)threadIdx.x; t < w; t += nthr)
                if (s_max[t + w] > s_max[t]) s_max[t] = s_max[t + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const double m = s_max[0];
            H->max_zeta_pi = m;
            S->k += 1;
            const int k = S->k;
            if (S->tr_dotr && k <= S->max_iter) {
                S->tr_alpha[k - 1] = phase == PH_SHP_END ? H->alpha_old : S->alpha;
                S->tr_omega[k - 1] = S->omega; S->tr_beta[k - 1] = S->beta; S->tr_dotr[k - 1] = S->dot_r;
            }
            // reference loop condition, src/shifted_solver.c:86 / 257 / 792
            if (!(m * m * S->dot_r > S->tol2 * S->dot_zero && k < S->max_iter)) S->done = 1;
            if (!(isfinite(S->alpha) && isfinite(S->beta) && isfinite(S->omega) && isfinite(S->dot_r)) && !S->breakdown_k)
                S->breakdown_k = k;
        }
    }
}



// u <- add + beta (u - omega w): daxpy(-omega) / dscal(beta) / daxpy(1.0)   (src/solver.c:217-219 etc.)
template <class T> __device__ __forceinline__ T recur3(T u, T w, T add, double omega, double beta)
{
    T t = u + (-omega) * w;
    t = beta * t;
    return t + 1.0 * add;
}

}  // namespace bicg
