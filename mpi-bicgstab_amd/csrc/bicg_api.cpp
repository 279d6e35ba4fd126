// bicg_api.cpp -- the additive handle API of include/bicgstab_hip.h: load / fetch / timed iteration / spmv / dot / spmm /
// shifted residuals / information calls. Split from bicg_solver.cpp in round 5; see bicg_host.h.
#include "bicg_host.h"

extern "C" {

int bicg_load(bicg_ctx *c, const double *x0, const double *b)
{
    use_device(c);
    x0 = host_in(c, x0); b = host_in(c, b);
    BICG_HIP(hipMemcpy(c->v.x, x0, sizeof(double) * c->n_loc, hipMemcpyHostToDevice));
    BICG_HIP(hipMemcpy(c->v.r, b, sizeof(double) * c->n_loc, hipMemcpyHostToDevice));
    return 0;
}

int bicg_fetch(bicg_ctx *c, double *x, double *r)
{
    use_device(c);
    BICG_HIP(hipStreamSynchronize(c->sc));
    x = host_out(c, x); r = host_out(c, r);
    if (x) BICG_HIP(hipMemcpy(x, c->v.x, sizeof(double) * c->n_loc, hipMemcpyDeviceToHost));
    if (r) BICG_HIP(hipMemcpy(r, c->v.r, sizeof(double) * c->n_loc, hipMemcpyDeviceToHost));
    return 0;
}

int bicg_run(bicg_ctx *c, int method, const bicg_options *opt, bicg_result *res) { return run_solver(c, method, opt, res); }
int bicg_run_begin(bicg_ctx *c, int method, const bicg_options *opt) { run_begin(c, method, opt); return 0; }
int bicg_run_iterate(bicg_ctx *c, int nsteps) { return run_iterate(c, nsteps); }
int bicg_run_iterate_timed(bicg_ctx *c, int nsteps, double ms[3])
{
    use_device(c);
    if (!c->region_ev[0]) for (auto &e : c->region_ev) BICG_HIP(hipEventCreate(&e));
    c->t_enq = 0.0;
    const double t0 = now_sec();
    BICG_HIP(hipEventRecord(c->region_ev[0], c->sc));
    const int k = run_iterate(c, nsteps);
    BICG_HIP(hipEventRecord(c->region_ev[1], c->sc));
    BICG_HIP(hipEventSynchronize(c->region_ev[1]));
    float dev = 0.f;
    BICG_HIP(hipEventElapsedTime(&dev, c->region_ev[0], c->region_ev[1]));
    ms[0] = dev; ms[1] = 1e3 * c->t_enq; ms[2] = 1e3 * (now_sec() - t0);
    return k;
}
int bicg_run_end(bicg_ctx *c, bicg_result *res) { return run_end(c, res); }
int bicg_sync(bicg_ctx *c)
{
    use_device(c);
    BICG_HIP(hipStreamSynchronize(c->sc));
    if (c->sm) BICG_HIP(hipStreamSynchronize(c->sm));
    return 0;
}

int bicg_solve(bicg_ctx *c, int method, double *x, double *r, const bicg_options *opt, bicg_result *res)
{
    bicg_load(c, x, r);
    const int k = run_solver(c, method, opt, res);
    bicg_fetch(c, x, r);
    return k;
}

int bicg_trace(bicg_ctx *c, double *alpha, double *omega, double *beta, double *dot_r)
{
    const int k = c->last_iters;
    if (k <= 0 || !c->trace) return 0;
    double *dst[4] = {alpha, omega, beta, dot_r};
    for (int i = 0; i < 4; ++i)
        if (dst[i]) BICG_HIP(hipMemcpy(dst[i], c->trace + (size_t)i * c->trace_cap, sizeof(double) * k, hipMemcpyDeviceToHost));
    return k;
}

static void reset_scal(bicg_ctx *c) { scal_reset(c); }

int bicg_spmv(bicg_ctx *c, const double *x, double *y)
{
    use_device(c);
    reset_scal(c);
    x = host_in(c, x); y = host_out(c, y);
    BICG_HIP(hipMemcpyAsync(c->v.p, x, sizeof(double) * c->n_loc, hipMemcpyHostToDevice, c->sc));
    c->time_kernels = false;
    spmv(c, c->v.p, c->v.s, 0, nullptr, c->red(0, PH_NONE));
    BICG_HIP(hipMemcpyAsync(y, c->v.s, sizeof(double) * c->n_loc, hipMemcpyDeviceToHost, c->sc));
    if (c->p2p) fetch_scal(c);      // also reports a peer that never delivered its halo values
    else BICG_HIP(hipStreamSynchronize(c->sc));
    return 0;
}

double bicg_dot(bicg_ctx *c, const double *x, const double *y)
{
    use_device(c);
    reset_scal(c);
    if (c->phantom) { x = host_in(c, x); y = x; }
    BICG_HIP(hipMemcpyAsync(c->v.p, x, sizeof(double) * c->n_loc, hipMemcpyHostToDevice, c->sc));
    BICG_HIP(hipMemcpyAsync(c->v.s, y, sizeof(double) * c->n_loc, hipMemcpyHostToDevice, c->sc));
    launch_dot(c->v.p, c->v.s, c->n_loc, c->S, c->red(0, PH_NONE, true, 1), c->sc);
    group_now(c, 1, PH_NONE);
    fetch_scal(c);
    return c->hS->red[0];
}

// Verification loop of the reference's shifted driver (src/test_shifted.c:129-154): for every shift the
// relative residual || (A + sigma_j I) x_j - b || / || b ||, computed on the device (SpMV with the
// shift folded into its epilogue + one fused difference/norm kernel per shift). Collective.
int bicg_shifted_residuals(bicg_ctx *c, const double *x_loc_set, const double *b_loc, const double *sigma, int nsig,
                           double *relres_out)
{
    use_device(c);
    reset_scal(c);
    x_loc_set = host_in(c, x_loc_set, (size_t)nsig); if (c->phantom) b_loc = x_loc_set;
    const size_t n = c->n_loc;
    BICG_HIP(hipMemcpyAsync(c->v.b, b_loc, sizeof(double) * n, hipMemcpyHostToDevice, c->sc));
    BICG_HIP(hipMemsetAsync(c->v.t, 0, sizeof(double) * c->stride, c->sc));
    c->time_kernels = false;
    launch_dot(c->v.b, c->v.b, c->n_loc, c->S, c->red(0, PH_NONE, true, 1), c->sc);
    group_now(c, 1, PH_NONE);
    fetch_scal(c);
    const double bb = c->hS->red[0];
    if (c->spmm_ok && !plan_off("spmm")) {
        // every matrix entry is read once for kSpmmCols shifts (the only place where
        // the reference multiplies A with many vectors is this verification loop, one SpMV per shift)
        spmm_buffers(c);
        std::vector<double> sq(kSpmmCols);
        for (int j0 = 0; j0 < nsig; j0 += kSpmmCols) {
            const int nv = std::min(kSpmmCols, nsig - j0);
            for (int j = 0; j < nv; ++j)
                BICG_HIP(hipMemcpyAsync(c->mm_in + (size_t)j * c->stride, x_loc_set + (size_t)(j0 + j) * n, sizeof(double) * n,
                                        hipMemcpyHostToDevice, c->sc));
            spmm_pass(c, nv, sigma + j0, true);
            BICG_HIP(hipMemcpyAsync(sq.data(), c->mm_out, sizeof(double) * kSpmmCols, hipMemcpyDeviceToHost, c->sc));
            fetch_scal(c);                                   // synchronises; reports a lost peer
            if (!c->single()) {                              // sum over ranks (host-side: kSpmmCols doubles)
                std::vector<int> cnt(c->nranks, (int)(sizeof(double) * kSpmmCols)), dsp(c->nranks);
                std::vector<double> all((size_t)c->nranks * kSpmmCols), mine((size_t)c->nranks * kSpmmCols);
                for (int p = 0; p < c->nranks; ++p) { dsp[p] = p * (int)(sizeof(double) * kSpmmCols); std::copy(sq.begin(), sq.end(), mine.begin() + (size_t)p * kSpmmCols); }
                c->comm->alltoallv_host(mine.data(), cnt.data(), dsp.data(), all.data(), cnt.data(), dsp.data());
                std::copy(sq.begin(), sq.end(), all.begin() + (size_t)c->rank * kSpmmCols);
                for (int j = 0; j < kSpmmCols; ++j) { double t = 0.0; for (int p = 0; p < c->nranks; ++p) t += all[(size_t)p * kSpmmCols + j]; sq[j] = t; }
            }
            for (int j = 0; j < nv; ++j) relres_out[j0 + j] = bb > 0.0 ? sqrt(sq[j] / bb) : sqrt(sq[j]);
        }
        return 0;
    }
    Vecs w = c->v;
    w.r = c->v.t;                               // zero vector: FDrift then yields || b - A x ||^2
    for (int j = 0; j < nsig; ++j) {
        BICG_HIP(hipMemcpyAsync(c->v.p, x_loc_set + (size_t)j * n, sizeof(double) * n, hipMemcpyHostToDevice, c->sc));
        c->cur_shift = sigma[j]; c->cur_has_shift = true;
        spmv(c, c->v.p, c->v.ax, 0, nullptr, c->red(0, PH_NONE));
        c->cur_has_shift = false; c->cur_shift = 0.0;
        launch_drift(w, Launch{c->S, Finish{}, c->sc}, c->red(0, PH_NONE, true, 2));
        group_now(c, 2, PH_NONE);
        fetch_scal(c);
        relres_out[j] = bb > 0.0 ? sqrt(c->hS->red[0] / bb) : sqrt(c->hS->red[0]);
    }
    return 0;
}

// Y_j = (A + sigma_j I) X_j, j < nvec, with A read once per kSpmmCols vectors ("batched SpMV", BASELINE.json configs[4]);
// x_loc_set / y_loc_set shift-major like the shifted solvers' x_loc_set; sigma may be NULL. Returns 1 (nothing done)
// when the matrix is not entirely on the sliced-ELL path. ms_out (optional): device time of the passes.
int bicg_spmm(bicg_ctx *c, const double *x_loc_set, const double *sigma, int nvec, double *y_loc_set, double *ms_out)
{
    use_device(c);
    if (!c->spmm_ok) return 1;
    reset_scal(c);
    spmm_buffers(c);
    std::vector<double> ph_y;
    if (c->phantom) { x_loc_set = host_in(c, x_loc_set, (size_t)nvec); ph_y.assign((size_t)nvec, 0.0); y_loc_set = ph_y.data(); }
    const size_t n = c->n_loc;
    hipEvent_t e0, e1;
    BICG_HIP(hipEventCreate(&e0)); BICG_HIP(hipEventCreate(&e1));
    float total = 0.f;
    for (int j0 = 0; j0 < nvec; j0 += kSpmmCols) {
        const int nv = std::min(kSpmmCols, nvec - j0);
        for (int j = 0; j < nv; ++j)
            BICG_HIP(hipMemcpyAsync(c->mm_in + (size_t)j * c->stride, x_loc_set + (size_t)(j0 + j) * n, sizeof(double) * n,
                                    hipMemcpyHostToDevice, c->sc));
        if (sigma) spmm_stage_sigma(c, nv, sigma + j0);
        // the events ride on the kernel's launch (stamped at its start and end, what rocprofv3 reports): events recorded around the
        // launch count the dispatch from an idle queue behind the copies as kernel time (181-184 us against 155)
        spmm_pass(c, nv, sigma ? sigma + j0 : nullptr, false, true, e0, e1);
        if (!c->mm_win) launch_vectors_from_rows(c->mm_yt, c->stride, nv, c->n_loc, c->mm_in, c->sc);     // result back to shift-major (reuses mm_in)
        const double *ysrc = c->mm_win ? c->mm_yt : c->mm_in;
        for (int j = 0; j < nv; ++j)
            BICG_HIP(hipMemcpyAsync(y_loc_set + (size_t)(j0 + j) * n, ysrc + (size_t)j * c->stride, sizeof(double) * n,
                                    hipMemcpyDeviceToHost, c->sc));
        fetch_scal(c);
        float ms = 0.f;
        BICG_HIP(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ms_out) *ms_out = (double)total;
    return 0;
}

int bicg_spmv_bench(bicg_ctx *c, int reps, double *ms_per_spmv)
{
    use_device(c);
    reset_scal(c);
    std::vector<double> ones(c->n_loc, 1.0);
    BICG_HIP(hipMemcpyAsync(c->v.p, ones.data(), sizeof(double) * c->n_loc, hipMemcpyHostToDevice, c->sc));
    BICG_HIP(hipStreamSynchronize(c->sc));
    c->time_kernels = false;
    for (int i = 0; i < 3; ++i) spmv(c, c->v.p, c->v.s, 0, nullptr, c->red(0, PH_NONE));
    hipEvent_t a, b;
    BICG_HIP(hipEventCreate(&a)); BICG_HIP(hipEventCreate(&b));
    BICG_HIP(hipEventRecord(a, c->sc));
    for (int i = 0; i < reps; ++i) spmv(c, c->v.p, c->v.s, 0, nullptr, c->red(0, PH_NONE));
    BICG_HIP(hipEventRecord(b, c->sc));
    BICG_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    BICG_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (c->p2p) fetch_scal(c);
    *ms_per_spmv = (double)ms / (reps > 0 ? reps : 1);
    return 0;
}

int bicg_comm_failed(bicg_ctx *c) { return c->comm_failed ? 1 : 0; }

int bicg_section_times(bicg_ctx *c, double ms[4], int *iterations, int *marks)
{
    if (!c) return 1;
    for (int i = 0; i < SEC_COUNT; ++i) ms[i] = c->sec_ms[i];
    if (iterations) *iterations = c->sec_iters;
    if (marks) *marks = c->sec_exhausted ? -c->sec_used : c->sec_used;
    return c->sec_used > 0 ? 0 : 2;
}

int bicg_plan_info(bicg_ctx *c, unsigned int out[8])
{
    out[0] = c->n_loc; out[1] = c->nnz_d; out[2] = c->nnz_o; out[3] = c->halo;
    out[4] = c->nblk + c->ng_int + c->ng_bnd;       // workgroups per SpMV
    out[5] = c->n_bnd + c->ng_bnd;                  // of which halo-touching
    out[6] = c->sell_rows;                          // rows on the sliced-ELL path
    out[7] = (unsigned)(c->sell_entries > c->sell_nnz ? c->sell_entries - c->sell_nnz : 0);   // padding entries
    return 0;
}

unsigned long long bicg_device_matrix_bytes(bicg_ctx *c) { return c->device_matrix_bytes; }
unsigned long long bicg_uniform_entries(bicg_ctx *c) { return c->uniform_entries; }
unsigned long long bicg_constant_entries(bicg_ctx *c) { return c->constant_entries; }
unsigned long long bicg_masked_rows(bicg_ctx *c) { return c->masked_rows; }
// out = {mailbox all-reduce p50, p99, hand-off wait p50, p99 (microseconds), samples of the former, of the latter}; returns 0 when
// the last solve recorded something (multi-rank persistent launches only)
int bicg_comm_wait_stats(bicg_ctx *c, double out[6])
{
    for (int i = 0; i < 6; ++i) out[i] = 0.0;
    if (!c->waitlog) return 1;
    use_device(c);
    std::vector<unsigned> h(3 * (size_t)kWaitCap);
    BICG_HIP(hipMemcpy(h.data(), c->waitlog, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost));
    auto pct = [](std::vector<unsigned> &v, double q) -> double {
        if (v.empty()) return 0.0;
        std::sort(v.begin(), v.end());
        return 0.01 * (double)v[std::min(v.size() - 1, (size_t)(q * (double)(v.size() - 1) + 0.5))];      // 100 MHz ticks -> us
    };
    std::vector<unsigned> mail, hand[2];
    for (size_t i = 0; i < kWaitCap; ++i) {
        if (h[i]) mail.push_back(h[i]);
        if (h[kWaitCap + i]) hand[0].push_back(h[kWaitCap + i]);
        if (h[2 * kWaitCap + i]) hand[1].push_back(h[2 * kWaitCap + i]);
    }
    // the row workgroup that borders another rank waits for halo values, the other one only for its own GPU: report the slower
    std::vector<unsigned> &hw = pct(hand[0], 0.5) >= pct(hand[1], 0.5) ? hand[0] : hand[1];
    out[0] = pct(mail, 0.5); out[1] = pct(mail, 0.99); out[2] = pct(hw, 0.5); out[3] = pct(hw, 0.99);
    out[4] = (double)mail.size(); out[5] = (double)hw.size();
    return mail.empty() && hw.empty() ? 1 : 0;
}
int bicg_stencil_info(bicg_ctx *c, unsigned int out[8])
{
    const bool on = stencil_product(c);
    const StencilDev &t = c->st;
    const unsigned int v[8] = {on ? 1u : 0u, t.sy, t.ny, t.nz, t.lines, t.zl, on ? stencil_grid(t) : 0u, t.nmc};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return on ? 1 : 0;
}
unsigned int bicg_stencil_rows_per_lane(bicg_ctx *c) { return stencil_product(c) ? (c->st.wide ? c->st.wide : 1u) : 0u; }
unsigned int bicg_plan_collisions(bicg_ctx *c) { return c->plan_collisions; }
unsigned int bicg_product_kernels(int reset)
{
    const unsigned m = g_product_kernels;
    if (reset) g_product_kernels = 0;
    return m;
}
unsigned long long bicg_spmv_matrix_bytes(bicg_ctx *c) { return stencil_product(c) ? c->stencil_matrix_bytes : c->matrix_bytes; }
int bicg_last_shifted_persistent(bicg_ctx *c) { return c->last_shifted_persist ? 1 : 0; }
int bicg_last_spmm_windowed(bicg_ctx *c) { return c->mm_win ? (c->mm_dma ? 2 : 1) : 0; }      // 2: the pipelined kernel (bicg_spmm.hip)

unsigned int bicg_ctx_flags(bicg_ctx *c)
{
    unsigned f = 0;
    if (c->p2p) f |= BICG_FLAG_P2P;
    if (c->ll_fused) f |= BICG_FLAG_LL_FUSED;
    if (c->overlap) f |= BICG_FLAG_OVERLAP;
    if (c->s_col16) f |= BICG_FLAG_COL16;
    if (c->sell_jag) f |= BICG_FLAG_JAGGED;
    if (c->win_slots) f |= BICG_FLAG_WINDOW;
    if (c->spmm_ok) f |= BICG_FLAG_SPMM;
    if (c->glist_all) f |= BICG_FLAG_ALL_SELL;
    if (c->rowsplit) f |= BICG_FLAG_ROWSPLIT;
    if (c->persist_on) f |= BICG_FLAG_PERSIST;
    if (c->fuse_pipe && c->fuse_plan_ok && !hosted(c)) f |= BICG_FLAG_FUSE_PIPE;
    if (c->pipe_probed && (c->probe_ms[0] > 0.0 || c->probe_ms[1] > 0.0)) f |= BICG_FLAG_PIPE_PROBED;
    if (c->uniform_entries) f |= BICG_FLAG_UNIFORM;
    if (c->constant_entries) f |= BICG_FLAG_CONSTANT;
    return f;
}

int bicg_solve_shifted(bicg_ctx *c, int variant, double *x_loc_set, double *r_loc, const double *sigma, int sigma_len,
                       int seed, const bicg_options *opt, bicg_result *res)
{
    return run_shifted(c, variant, x_loc_set, r_loc, sigma, sigma_len, seed, opt, res);
}


}  // extern "C"
