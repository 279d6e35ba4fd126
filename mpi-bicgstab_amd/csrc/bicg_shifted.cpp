// bicg_shifted.cpp -- the shifted family on the device: shifted_lopbicgstab / shifted_pipe_lopbicgstab / shifted_bicgstab
// (reference src/shifted_solver.c) and shifted_lopbicg / shifted_lopbicg_switching (src/shifted_switching_solver.c), with the
// reference's section prints. Split from bicg_solver.cpp in round 5; see bicg_host.h.
#include "bicg_host.h"

// ---------------------------------------------------------------- shifted BiCGStab
// (A + sigma_j I) x_j = b for all j from ONE Krylov recurrence on the seed system: 2 SpMV per
// iteration whatever the number of shifts (reference src/shifted_solver.c:182-354). Per iteration:
// SpMV(+sigma_seed) with (r#,s) | q, r_old | SpMV(+sigma_seed) with (q,y),(q,q) | ONE batched kernel
// over all shifts (x_seed, r, every p_j and x_j, two dots) | p_seed.  The per-shift scalar
// recurrences (beta_j, pi_j, eta_j, alpha_j, omega_j, zeta_j) run on the device, one thread per shift.
// shifted_lopbicg / shifted_lopbicg_switching (+_noovlp), reference src/shifted_switching_solver.c.
// Per iteration: SpMV (+alpha) ; q ; SpMV (+omega) ; seed update with the (r,r), (r#,r) dots (+beta
// and every active shift's coefficients) ; ONE batched kernel over all shifts ; a one-workgroup
// kernel for the stop flags. A seed switch needs new vector pointers and a rescaled r from the
// host, so the device raises done/paused, the launches already queued fall through, and the host
// resumes with the new seed (switches are rare: at most one per shift).
// "Seed time" / "Shift time" as the reference prints them under MEASURE_SECTION_TIME (src/shifted_solver.c:244-247,
// src/shifted_switching_solver.c:563-...): shift = the passes over the shifted systems, seed = total - shift
void print_sections(const bicg_ctx *c, double total_seconds)
{
    if (c->sec_used == 0) return;
    const double shift = c->sec_ms[SEC_SHIFT] * 1.0e-3;
    printf("Seed time    : %e [sec.]\n", total_seconds - shift);
    printf("Shift time   : %e [sec.]\n", shift);
}

// BICG_SECTION_TIME=2 (bicg_options.time_kernels & 4) on the switching solvers: the reference's DISPLAY_SECTION_TIME table
// (src/shifted_switching_solver.c:884-892: one line per iteration) and the ten totals it prints at the end (:994-1005), on the
// device clock. Mapping: agv = halo pack + exchange (host / RCCL transports; with the peer-to-peer path the exchange is inside the
// product's launch and shows under mult_diag), mult_diag = the rows without halo entries (one rank: every row), mult_offd = the
// halo-touching rows (their diag AND offd part: one kernel), ared = the hand-over of the dot groups, shift = the batched pass
// over the shifted systems, seed = everything of the iteration except shift, switch = host time spent in seed switches.
void print_section_table(bicg_ctx *c, int its, int nsig, const int *unsolved, double total_seconds)
{
    if (c->sec_used == 0 || its <= 0) return;
    enum { AGV1, DIAG1, OFFD1, AGV2, DIAG2, OFFD2, ARED, SHIFT, SEED, NCOL };
    std::vector<double> t((size_t)(its + 1) * NCOL, 0.0);
    for (int i = 0; i + 1 < c->sec_used; ++i) {
        if (c->sec_lab[i] == SEC_STOP) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->sec_ev[i], c->sec_ev[i + 1]) != hipSuccess) continue;
        const int k = std::min(std::max(c->sec_k[i], 0), its), prod = c->sec_sub[i] >> 4, sub = c->sec_sub[i] & 15;
        double *row = t.data() + (size_t)k * NCOL;
        const double sec = 1.0e-3 * ms;
        if (c->sec_lab[i] == SEC_SHIFT) { row[SHIFT] += sec; continue; }
        row[SEED] += sec;
        if (c->sec_lab[i] == SEC_REDUCE) row[ARED] += sec;
        else if (c->sec_lab[i] == SEC_SPMV && (prod == 1 || prod == 2)) row[(prod == 1 ? AGV1 : AGV2) + (sub == 1 ? 0 : sub == 2 ? 2 : 1)] += sec;
    }
    printf("iter, unsolved, seed, agv_1, mult_diag_1, mult_offd_1, agv_2, mult_diag_2, mult_offd_2, ared, shift\n");
    double tot[NCOL] = {0};
    for (int k = 1; k <= its; ++k) {
        const double *r = t.data() + (size_t)k * NCOL;
        printf("%d, %d, %e, %e, %e, %e, %e, %e, %e, %e, %e\n", k, unsolved ? unsolved[k] : nsig, r[SEED], r[AGV1], r[DIAG1], r[OFFD1], r[AGV2], r[DIAG2],
               r[OFFD2], r[ARED], r[SHIFT]);
        for (int q = 0; q < NCOL; ++q) tot[q] += r[q];
    }
    printf("Seed time    : %e [sec.]\n", total_seconds - tot[SHIFT] - c->switch_sec);
    printf(" 1 Agv time   : %e [sec.]\n", tot[AGV1]);
    printf(" 1 Mult_diag  : %e [sec.]\n", tot[DIAG1]);
    printf(" 1 Mult_offd  : %e [sec.]\n", tot[OFFD1]);
    printf(" 2 Agv time   : %e [sec.]\n", tot[AGV2]);
    printf(" 2 Mult_diag  : %e [sec.]\n", tot[DIAG2]);
    printf(" 2 Mult_offd  : %e [sec.]\n", tot[OFFD2]);
    printf(" Ared time    : %e [sec.]\n", tot[ARED]);
    printf("Shift time   : %e [sec.]\n", tot[SHIFT]);
    printf("Switch time  : %e [sec.]\n", c->switch_sec);
}

int run_switching(bicg_ctx *c, int mode, double *x_set_host, double *r_host, const double *sigma, int nsig, int seed,
                  const bicg_options *opt_in, bicg_result *res)
{
    std::vector<double> ph_x, ph_r;      // a rank without rows: the caller's vectors are empty (bicg_ctx::phantom)
    if (c->phantom && nsig > 0) { ph_x.assign((size_t)nsig, 0.0); ph_r.assign(1, 0.0); x_set_host = ph_x.data(); r_host = ph_r.data(); }
    bicg_options o;
    if (opt_in) o = *opt_in; else { bicg_default_options(&o); o.tol = 1.0e-12; }   // EPS of src/shifted_switching_solver.c:5
    if (nsig < 1 || seed < 0 || seed >= nsig) die("bicg_solve_shifted", "seed outside the shift list");
    if (o.max_iter < 0) o.max_iter = 0;
    if (o.check_every < 1) o.check_every = 1;
    use_device(c);
    c->wave_mode = false;                // the shifted solvers keep the ticket reductions (scalars applied in place)
    c->spmv_dir = 0;                     // same first direction for every solve on this context (see run_begin)
    const size_t st = c->stride, n = c->n_loc;

    if (c->sh_cap < nsig) {
        for (void *p : {(void *)c->sh_dev, (void *)c->sh_arrays, (void *)c->p_set, (void *)c->x_set}) if (p) BICG_HIP(hipFree(p));
        c->sh_dev = dev_alloc<ShiftDev>(1);
        c->sh_arrays = dev_alloc<double>(12 * (size_t)nsig);
        c->p_set = dev_alloc<double>((size_t)nsig * st);
        c->x_set = dev_alloc<double>((size_t)nsig * st);
        c->sh_cap = nsig;
    }
    const int L = o.max_iter + 2;                                    // archive entries 0 .. max_iter + 1
    const size_t nd = 3 * (size_t)L + (size_t)nsig * L, ni = 2 * (size_t)nsig + (size_t)L;     // (+ the systems still running, per iteration)
    const size_t need = nd * sizeof(double) + ni * sizeof(int);
    if (c->sw_cap < need) {
        if (c->sw_buf) BICG_HIP(hipFree(c->sw_buf));
        BICG_HIP(hipMalloc((void **)&c->sw_buf, need));
        c->sw_cap = need;
    }
    ShiftDev h;
    memset(&h, 0, sizeof h);
    h.nsig = nsig; h.seed = seed; h.mode = mode; h.arc_len = L;
    double **arr[12] = {&h.sigma, &h.alpha, &h.beta, &h.omega, &h.eta, &h.zeta, &h.pi_old, &h.pi_new, &h.cp, &h.cx, &h.c1, &h.c2};
    for (int i = 0; i < 12; ++i) *arr[i] = c->sh_arrays + (size_t)i * nsig;
    h.a_arc = c->sw_buf; h.b_arc = h.a_arc + L; h.w_arc = h.b_arc + L; h.pi_arc = h.w_arc + L;
    h.stop = (int *)(c->sw_buf + nd); h.skip = h.stop + nsig; h.unsolved_arc = h.skip + nsig;
    BICG_HIP(hipMemcpy(c->sh_dev, &h, sizeof h, hipMemcpyHostToDevice));
    BICG_HIP(hipMemset(c->sh_arrays, 0, sizeof(double) * 12 * (size_t)nsig));
    BICG_HIP(hipMemset(c->sw_buf, 0, need));
    BICG_HIP(hipMemcpy(h.sigma, sigma, sizeof(double) * nsig, hipMemcpyHostToDevice));
    BICG_HIP(hipMemset(c->p_set, 0, sizeof(double) * (size_t)nsig * st));
    BICG_HIP(hipMemset(c->x_set, 0, sizeof(double) * (size_t)nsig * st));
    for (int j = 0; j < nsig; ++j)
        BICG_HIP(hipMemcpy(c->x_set + (size_t)j * st, x_set_host + (size_t)j * n, sizeof(double) * n, hipMemcpyHostToDevice));
    BICG_HIP(hipMemcpy(c->v.r, r_host, sizeof(double) * n, hipMemcpyHostToDevice));
    BICG_HIP(hipDeviceSynchronize());

    if (c->trace_cap < o.max_iter) {
        if (c->trace) BICG_HIP(hipFree(c->trace));
        c->trace_cap = o.max_iter > 0 ? o.max_iter : 1;
        c->trace = dev_alloc<double>(4 * (size_t)c->trace_cap);
    }
    Scal hs;
    memset(&hs, 0, sizeof hs);
    hs.tol2 = o.tol * o.tol; hs.max_iter = o.max_iter;
    hs.tr_alpha = c->trace; hs.tr_omega = c->trace + c->trace_cap;
    hs.tr_beta = c->trace + 2 * (size_t)c->trace_cap; hs.tr_dotr = c->trace + 3 * (size_t)c->trace_cap;
    hs.sh = c->sh_dev;
    BICG_HIP(hipMemcpyAsync(c->S, &hs, sizeof hs, hipMemcpyHostToDevice, c->sc));
    BICG_HIP(hipMemsetAsync(c->counter, 0, sizeof(unsigned) * (kShards + 1) * kCounterStride, c->sc));
    BICG_HIP(hipMemsetAsync(c->slab + 2 * st, 0, sizeof(double) * 10 * st, c->sc));
    c->time_kernels = false;
    sec_begin(c, (o.time_kernels & 2) != 0); c->sec_dump = (o.time_kernels & 4) != 0;
    for (int j = 0; j < nsig; ++j)          // p[sigma] <- b for EVERY shift, src/shifted_switching_solver.c:348
        BICG_HIP(hipMemcpyAsync(c->p_set + (size_t)j * st, c->v.r, sizeof(double) * n, hipMemcpyDeviceToDevice, c->sc));
    {   // streaming policy: matrix + 7 work vectors + the two sets
        // (the two sets are streamed past the cache by their own kernels: what competes with the matrix for it are the work vectors.
        // 16 shifts, Transport-shaped: 250 against 276 us per iteration with ordinary loads -- profiles/r05/ab_matrix_stream_policy.txt)
        const double ws = (double)c->matrix_bytes + 8.0 * st * 7;
        c->sell_nt = ws > 2.5 * 256.0 * 1048576.0;
        if (c->sell_nt_env >= 0) c->sell_nt = c->sell_nt_env != 0;
    }
    BICG_HIP(hipStreamSynchronize(c->sc));

    Vecs &v = c->v;
    double *qc = v.z;                               // q_copy (:394)
    const double t0 = now_sec();
    c->cur_has_shift = false;
    launch_shift_init(v, c->p_set + (size_t)seed * st, c->S, c->red(0, PH_SW_INIT, true, 1), c->sc);   // r# = r, (r,r)
    group_now(c, 1, PH_SW_INIT);
    c->cur_has_shift = true;
    int switches = 0;
    for (;;) {
        fetch_scal(c);
        if (c->hS->paused) {                        // a seed switch happened at the end of iteration hS->k
            const double t_sw = now_sec();
            ShiftDev now;
            BICG_HIP(hipMemcpy(&now, c->sh_dev, sizeof now, hipMemcpyDeviceToHost));
            launch_scale(v.r, (uint32_t)n, now.r_scale, c->sc);                       // (:499)
            seed = now.seed;
            ++switches;
            const bool finished = c->hS->paused == 2;
            // the reference's line at every switch (src/shifted_switching_solver.c:526; its k counts from 1). Its per-shift
            // "sigma[j] eta: ..." debug lines (:522) are not reproduced.
            if (c->rank == 0 && !o.quiet && !finished)
                printf("k: %d, seed: %d, remain: %d\n", c->hS->k + 1, seed, nsig - now.stop_count);
            const int zero2[2] = {0, 0};
            if (!finished) BICG_HIP(hipMemcpyAsync(&c->S->done, &zero2[0], sizeof(int), hipMemcpyHostToDevice, c->sc));
            BICG_HIP(hipMemcpyAsync(&c->S->paused, &zero2[1], sizeof(int), hipMemcpyHostToDevice, c->sc));
            BICG_HIP(hipStreamSynchronize(c->sc));
            c->switch_sec += now_sec() - t_sw;      // the reference's switch_time (src/shifted_switching_solver.c:488-530)
            if (finished) { c->hS->paused = 0; break; }
            continue;
        }
        if (c->hS->done || c->hS->k >= o.max_iter) break;
        double *p_seed = c->p_set + (size_t)seed * st, *x_seed = c->x_set + (size_t)seed * st;
        c->cur_shift = sigma[seed];
        const int chunk = std::min(o.check_every, o.max_iter - c->hS->k);
        sec_mark(c, SEC_VEC);
        for (int j = 0; j < chunk; ++j) {
            c->cur_k = c->hS->k + j + 1; c->cur_prod = 1; sec_remark(c);
            spmv(c, p_seed, v.s, 1, v.rh, c->red(0, PH_SW_ALPHA, true, 1));           // s = (A + sigma I) p[seed], (r#,s)
            group_now(c, 1, PH_SW_ALPHA);
            launch_sw_q(v, qc, c->S, c->sc);                                          // r_old, q
            c->cur_prod = 2;
            spmv(c, v.r, v.y, 3, v.r, c->red(0, PH_SW_OMEGA, true, 2));               // y = (A + sigma I) q, (q,y), (q,q)
            c->cur_prod = 0;
            group_now(c, 2, PH_SW_OMEGA);
            launch_sw_seed(v, x_seed, p_seed, c->S, c->red(0, PH_SW_END, true, 2), c->sc);   // x[seed], r, (r,r), (r#,r)
            group_now(c, 2, PH_SW_END);
            {
                Section sec(c, SEC_SHIFT);       // the shift loops of src/shifted_switching_solver.c:425-480
                launch_sw_shifts(v, qc, c->p_set, c->x_set, (uint32_t)st, seed, c->sh_dev, c->S, c->sc);
            }
            launch_apply(c->S, PH_SW_STOP, c->sc);                                    // identical on every rank: no sums
        }
        sec_mark(c, SEC_STOP);
    }
    c->cur_has_shift = false; c->cur_shift = 0.0;
    const double t1 = now_sec();

    const int its = c->hS->k;
    c->last_iters = its;
    sec_collect(c, its);
    for (int j = 0; j < nsig; ++j)
        BICG_HIP(hipMemcpy(x_set_host + (size_t)j * n, c->x_set + (size_t)j * st, sizeof(double) * n, hipMemcpyDeviceToHost));
    BICG_HIP(hipMemcpy(r_host, c->v.r, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (res) {
        memset(res, 0, sizeof *res);
        res->iterations = its; res->dot_r = c->hS->dot_r; res->dot_zero = c->hS->dot_zero;
        res->seconds = t1 - t0; res->iter_seconds = t1 - t0;
        res->breakdown_iteration = c->hS->breakdown_k;
        res->adaptive_replacements = switches;      // reused: number of seed switches
    }
    const int k_ref = mode == SH_SWITCH ? its + 1 : its;   // the switching variants count from 1 (:295, 536)
    if (c->rank == 0 && !o.quiet) {   // reference src/shifted_switching_solver.c:228-233 / :556-560
        if (mode == SH_SWITCH) printf("Total iter   : %d\n", k_ref - 1);
        printf("Total time   : %e [sec.] \n", t1 - t0);
        printf("Avg time/iter: %e [sec.] \n", (t1 - t0) / (k_ref > 0 ? k_ref : 1));
        if (c->sec_dump && c->sec_used > 0) {
            std::vector<int> unsolved((size_t)L, nsig);
            BICG_HIP(hipMemcpy(unsolved.data(), h.unsolved_arc, sizeof(int) * (size_t)L, hipMemcpyDeviceToHost));
            print_section_table(c, its, nsig, unsolved.data(), t1 - t0);
        } else {
            print_sections(c, t1 - t0);
            if (c->sec_used > 0 && mode == SH_SWITCH) printf("Switch time  : %e [sec.]\n", c->switch_sec);      // (src/shifted_switching_solver.c:566)
        }
        fflush(stdout);
    }
    return k_ref;
}

int run_shifted(bicg_ctx *c, int mode, double *x_set_host, double *r_host, const double *sigma, int nsig, int seed,
                const bicg_options *opt_in, bicg_result *res)
{
    std::vector<double> ph_x, ph_r;      // a rank without rows: the caller's vectors are empty (bicg_ctx::phantom)
    if (c->phantom && nsig > 0) { ph_x.assign((size_t)nsig, 0.0); ph_r.assign(1, 0.0); x_set_host = ph_x.data(); r_host = ph_r.data(); }
    if (mode == SH_FLAG || mode == SH_SWITCH) return run_switching(c, mode, x_set_host, r_host, sigma, nsig, seed, opt_in, res);
    if (mode < SH_LOP || mode > SH_XI) die("bicg_solve_shifted", "unknown variant");
    if (mode == SH_XI) seed = 0;          // shifted_bicgstab: the seed system is A itself, shift index 0
    bicg_options o;
    if (opt_in) o = *opt_in; else { bicg_default_options(&o); o.tol = 1.0e-12; }   // EPS of src/shifted_solver.c:5
    if (nsig < 1 || seed < 0 || seed >= nsig) die("bicg_solve_shifted", "seed outside the shift list");
    if (o.max_iter < 0) o.max_iter = 0;
    if (o.check_every < 1) o.check_every = 1;
    use_device(c);
    c->wave_mode = false;                // the shifted solvers keep the ticket reductions (scalars applied in place)
    c->spmv_dir = 0;                     // same first direction for every solve on this context (see run_begin)
    const size_t st = c->stride, n = c->n_loc;

    if (c->sh_cap < nsig) {
        for (void *p : {(void *)c->sh_dev, (void *)c->sh_arrays, (void *)c->p_set, (void *)c->x_set}) if (p) BICG_HIP(hipFree(p));
        c->sh_dev = dev_alloc<ShiftDev>(1);
        c->sh_arrays = dev_alloc<double>(12 * (size_t)nsig);
        c->p_set = dev_alloc<double>((size_t)nsig * st);
        c->x_set = dev_alloc<double>((size_t)nsig * st);
        c->sh_cap = nsig;
    }
    ShiftDev h;
    memset(&h, 0, sizeof h);
    h.nsig = nsig; h.seed = seed; h.mode = mode;
    double **arr[12] = {&h.sigma, &h.alpha, &h.beta, &h.omega, &h.eta, &h.zeta, &h.pi_old, &h.pi_new, &h.cp, &h.cx, &h.c1, &h.c2};
    for (int i = 0; i < 12; ++i) *arr[i] = c->sh_arrays + (size_t)i * nsig;
    BICG_HIP(hipMemcpy(c->sh_dev, &h, sizeof h, hipMemcpyHostToDevice));
    BICG_HIP(hipMemset(c->sh_arrays, 0, sizeof(double) * 12 * (size_t)nsig));
    BICG_HIP(hipMemcpy(h.sigma, sigma, sizeof(double) * nsig, hipMemcpyHostToDevice));
    BICG_HIP(hipMemset(c->p_set, 0, sizeof(double) * (size_t)nsig * st));        // calloc, src/shifted_solver.c:223
    BICG_HIP(hipMemset(c->x_set, 0, sizeof(double) * (size_t)nsig * st));
    for (int j = 0; j < nsig; ++j)
        BICG_HIP(hipMemcpy(c->x_set + (size_t)j * st, x_set_host + (size_t)j * n, sizeof(double) * n, hipMemcpyHostToDevice));
    BICG_HIP(hipMemcpy(c->v.r, r_host, sizeof(double) * n, hipMemcpyHostToDevice));
    BICG_HIP(hipDeviceSynchronize());       // the memsets above ran on the null stream; sc does not wait for it

    if (c->trace_cap < o.max_iter) {
        if (c->trace) BICG_HIP(hipFree(c->trace));
        c->trace_cap = o.max_iter > 0 ? o.max_iter : 1;
        c->trace = dev_alloc<double>(4 * (size_t)c->trace_cap);
    }
    Scal hs;
    memset(&hs, 0, sizeof hs);
    hs.tol2 = o.tol * o.tol; hs.max_iter = o.max_iter;
    hs.tr_alpha = c->trace; hs.tr_omega = c->trace + c->trace_cap;
    hs.tr_beta = c->trace + 2 * (size_t)c->trace_cap; hs.tr_dotr = c->trace + 3 * (size_t)c->trace_cap;
    hs.sh = c->sh_dev;
    BICG_HIP(hipMemcpyAsync(c->S, &hs, sizeof hs, hipMemcpyHostToDevice, c->sc));
    BICG_HIP(hipMemsetAsync(c->counter, 0, sizeof(unsigned) * (kShards + 1) * kCounterStride, c->sc));
    BICG_HIP(hipMemsetAsync(c->slab + 2 * st, 0, sizeof(double) * 10 * st, c->sc));
    c->time_kernels = false;
    sec_begin(c, (o.time_kernels & 2) != 0); c->sec_dump = (o.time_kernels & 4) != 0;
    if (mode == SH_XI)                      // p[sigma] <- b for every shift, src/shifted_solver.c:72
        for (int j = 0; j < nsig; ++j)
            BICG_HIP(hipMemcpyAsync(c->p_set + (size_t)j * st, c->v.r, sizeof(double) * n, hipMemcpyDeviceToDevice, c->sc));
    {   // streaming policy: matrix + 6 work vectors + the two sets
        // (as in run_switching: the sets are streamed, the work vectors compete with the matrix)
        const double ws = (double)c->matrix_bytes + 8.0 * st * (mode == SH_PIPE ? 10 : 6);
        c->sell_nt = ws > 2.5 * 256.0 * 1048576.0;
        if (c->sell_nt_env >= 0) c->sell_nt = c->sell_nt_env != 0;
    }
    BICG_HIP(hipStreamSynchronize(c->sc));

    double *p_seed = c->p_set + (size_t)seed * st;
    Vecs &v = c->v;
    const bool shifted_A = mode != SH_XI;       // lop / pipe iterate on A + sigma[seed] I, shifted_bicgstab on A
    const double t0 = now_sec();
    c->cur_has_shift = false;
    launch_shift_init(v, p_seed, c->S, c->red(0, PH_SH_INIT, true, 1), c->sc);
    group_now(c, 1, PH_SH_INIT);
    c->cur_shift = sigma[seed]; c->cur_has_shift = shifted_A;
    if (mode == SH_PIPE) {                                                   // src/shifted_solver.c:764-769, 785-786
        spmv(c, v.r, v.w, 1, v.r, c->red(0, PH_SHP_INIT_ALPHA));             // w = (A + sigma I) r, (r,w)
        group_defer(c, 1, PH_SHP_INIT_ALPHA);
        spmv(c, v.w, v.t, 0, nullptr, c->red(0, PH_NONE));                   // t = (A + sigma I) w
        group_flush(c);
    }
    fetch_scal(c);
    int it = 0;
    // latency-bound ranks: the pipelined shifted iteration as ONE persistent launch per chunk (bicg_persist.hip, k_shpipe_persist);
    // section timing needs the launch boundaries and keeps the multi-launch form
    const int persist_shifted_env = knob_tok("BICG_PERSIST", "shifted") ? atoi(knob_tok("BICG_PERSIST", "shifted")) : 1;
    bool persist = (mode == SH_PIPE || mode == SH_LOP) && c->persist_on && c->persist.rpt == 1u && nsig <= kPersistMaxShifts && persist_shifted_env != 0 &&
                   !(o.time_kernels & 3) && !c->time_sections;
    // BICG_DISPLAY_RESIDUAL=1: the progress line the reference prints when it is compiled with -DDISPLAY_RESIDUAL
    // (src/shifted_solver.c:151-155, 325-329, 500-504, 672-676, 870-874, 1061-1065: every OUT_ITER iterations the relative
    // residual and the largest |xi tau| / |1 / (zeta pi)| over the shifts). The host looks at the scalars every out_iter
    // iterations then (the largest ratio lives in the device's ShiftDev), in the multi-launch form.
    const char *show_env = getenv("BICG_DISPLAY_RESIDUAL");
    const bool show = show_env && atoi(show_env) != 0 && o.out_iter > 0;
    if (show) { o.check_every = o.out_iter; persist = false; }
    c->last_shifted_persist = false;
    while (!c->hS->done && it < o.max_iter) {
        const int persist_chunk_min = knob_tok("BICG_PERSIST", "chunk") ? std::max(1, atoi(knob_tok("BICG_PERSIST", "chunk"))) : kPersistChunk;
        const int chunk = std::min(persist ? std::max(o.check_every, persist_chunk_min) : o.check_every, o.max_iter - it);
        sec_mark(c, SEC_VEC);
        if (persist) {
            persist = persist_chunk_shifted(c, mode, chunk, it, nsig, seed, sigma[seed]);
            if (persist) c->last_shifted_persist = true;
        }
        for (int j = 0; j < chunk && !persist; ++j) {
            if (mode == SH_PIPE) {
                launch_shift_pipe1(v, p_seed, c->S, c->red(0, PH_SHP_OMEGA), c->sc);    // p, s, z, r_old, q, y, 2 dots
                group_defer(c, 2, PH_SHP_OMEGA);
                spmv(c, v.z, v.v, 0, nullptr, c->red(0, PH_NONE));                      // v = (A + sigma I) z
                {   // the shift loops of src/shifted_solver.c:850-905, with the seed system's x / r / w and the five dots in the same pass
                    Section sec(c, SEC_SHIFT);
                    launch_shift_pipe2(v, c->p_set, c->x_set, (uint32_t)st, seed, c->sh_dev, c->S, c->red(0, PH_SHP_END), c->sc);
                }
                group_defer(c, 5, PH_SHP_END);
                spmv(c, v.w, v.t, 0, nullptr, c->red(0, PH_NONE));                      // t = (A + sigma I) w
                group_flush(c);
                continue;
            }
            spmv(c, p_seed, v.s, 1, v.rh, c->red(0, PH_SH_ALPHA, true, 1));          // s = (A [+ sigma I]) p[seed], (r#,s)
            group_now(c, 1, PH_SH_ALPHA);
            launch_shift_q(v, c->S, c->sc);                                 // r_old = r, q = r - alpha s
            // lop: (q,y), (q,q) ; shifted_bicgstab: (q,y), (y,y)
            spmv(c, v.r, v.y, mode == SH_XI ? 2 : 3, v.r, c->red(0, PH_SH_OMEGA, true, 2));
            group_now(c, 2, PH_SH_OMEGA);
            {   // the shift loops of src/shifted_solver.c:132-154 and 180-208, with the seed system's x / r and two dots in the same pass
                Section sec(c, SEC_SHIFT);
                launch_shift_update(v, c->p_set, c->x_set, (uint32_t)st, seed, c->sh_dev, c->S, c->red(0, PH_SH_END, true, 2), c->sc);
            }
            group_now(c, 2, PH_SH_END);
            launch_shift_pseed(v, p_seed, c->S, c->sc);                     // p[seed] = r + beta (p[seed] - omega s)
        }
        it += chunk;
        sec_mark(c, SEC_STOP);
        fetch_scal(c);
        if (persist && mode == SH_PIPE) persist_account(c);
        if (show && c->hS->k == it && it % o.out_iter == 0) {       // (a solve that stopped inside the chunk has printed its last line)
            ShiftDev now;
            BICG_HIP(hipMemcpy(&now, c->sh_dev, sizeof now, hipMemcpyDeviceToHost));
            if (c->rank == 0 && !o.quiet)
                printf("Iteration: %d, Residual: %e, %s: %e\n", it, sqrt(c->hS->dot_r / c->hS->dot_zero), mode == SH_XI ? "Max_Xi" : "Max_Zeta_Pi",
                       now.max_zeta_pi);
        }
    }
    c->cur_has_shift = false; c->cur_shift = 0.0;
    const double t1 = now_sec();

    const int k = c->hS->k;
    c->last_iters = k;
    sec_collect(c, k);
    for (int j = 0; j < nsig; ++j)
        BICG_HIP(hipMemcpy(x_set_host + (size_t)j * n, c->x_set + (size_t)j * st, sizeof(double) * n, hipMemcpyDeviceToHost));
    BICG_HIP(hipMemcpy(r_host, c->v.r, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (res) {
        memset(res, 0, sizeof *res);
        res->iterations = k; res->dot_r = c->hS->dot_r; res->dot_zero = c->hS->dot_zero;
        res->seconds = t1 - t0; res->iter_seconds = t1 - t0;
    }
    if (c->rank == 0 && !o.quiet) {   // reference src/shifted_solver.c:336-343
        printf("Total iter   : %d\n", k);
        printf("Final r      : %e\n", sqrt(c->hS->dot_r / c->hS->dot_zero));
        printf("Total time   : %e [sec.] \n", t1 - t0);
        printf("Avg time/iter: %e [sec.] \n", (t1 - t0) / k);
        print_sections(c, t1 - t0);
        fflush(stdout);
    }
    return k;
}

