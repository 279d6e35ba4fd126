// bicg_dropin.cpp -- the reference's own call surface on top of the context API: bicgstab / ca_bicgstab / pipe_bicgstab /
// pipe_bicgstab_rr (src/solver.h:10-13), the shifted and seed-switching solvers (src/shifted_solver.h:16-21,
// src/shifted_switching_solver.h:10-12), the context that stays resident between calls, run-time options from the environment.
// Split from bicg_solver.cpp in round 5; see bicg_host.h.
#include "bicg_host.h"

void check_square(const INFO_Matrix *info)
{
    if (info->cols != info->rows) {   // reference src/solver.c:43-46
        printf("Error: matrix is not square.\n");
        exit(1);
    }
}

void env_options(bicg_options *o)
{
    bicg_default_options(o);
    if (const char *s = getenv("BICG_TOL")) o->tol = atof(s);
    if (const char *s = getenv("BICG_MAX_ITER")) o->max_iter = atoi(s);
    if (const char *s = getenv("BICG_OUT_ITER")) o->out_iter = atoi(s);
    if (const char *s = getenv("BICG_CHECK_EVERY")) o->check_every = atoi(s);
    if (const char *s = getenv("BICG_QUIET")) o->quiet = atoi(s);
    if (const char *s = getenv("BICG_RR_DRIFT")) o->rr_drift = atof(s);
    // the reference's MEASURE_SECTION_TIME (1) and DISPLAY_SECTION_TIME (2: the per-iteration table of the switching solvers)
    if (const char *s = getenv("BICG_SECTION_TIME")) o->time_kernels = atoi(s) >= 2 ? 6 : atoi(s) ? 2 : 0;
}

// ---------------------------------------------------------------- matrix residency across drop-in calls
// The reference's drivers call a solver many times on the same blocks (main_repeat.c:109-132: 10 x,
// main_seed_diff.c: 28 x); building the SpMV plan and uploading ~700 MB per call would cost more than
// the solves. The context of the last drop-in call stays resident and is reused when the caller
// passes the same blocks again: same array addresses, sizes and partition, AND the same contents --
// every value, column and row pointer goes through a 64-bit hash (one pass at memory speed, ~20 ms per
// 200 MB against ~1 s for plan + upload), because the caller may have edited the matrix in place
// between calls (the reference's csr_shift_diagonal does, src/matrix.c:518-531). Hit or miss is agreed
// by all ranks (bicg_create is collective). BICG_DROPIN_CACHE=0 restores create / destroy per call.
struct DropinKey {
    const void *dv, *dc, *dp, *ov, *oc, *op;
    unsigned rows, nnz_d, nnz_o, n_glob;
    int nranks, rank, first_row;
    const Comm *comm;
    const void *p2p;
    uint64_t hash;
    bool operator==(const DropinKey &o) const
    {
        return dv == o.dv && dc == o.dc && dp == o.dp && ov == o.ov && oc == o.oc && op == o.op && rows == o.rows &&
               nnz_d == o.nnz_d && nnz_o == o.nnz_o && n_glob == o.n_glob && nranks == o.nranks && rank == o.rank &&
               first_row == o.first_row && comm == o.comm && p2p == o.p2p && hash == o.hash;
    }
};
struct DropinCache { bicg_ctx *ctx = nullptr; DropinKey key{}; unsigned hits = 0, misses = 0; } g_dropin;

uint64_t hash_words(uint64_t h, const void *data, size_t bytes)
{
    // four independent multiply-xor lanes over 8-byte words: runs at memory speed, order-sensitive
    const uint64_t *w = (const uint64_t *)data;
    const size_t n = bytes / 8;
    uint64_t a = h ^ 0x9E3779B97F4A7C15ull, b = h + 0xBF58476D1CE4E5B9ull, c = ~h, d = h * 0x94D049BB133111EBull + 1;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        a = (a ^ w[i]) * 0x100000001B3ull; b = (b ^ w[i + 1]) * 0x9E3779B97F4A7C15ull;
        c = (c ^ w[i + 2]) * 0xC2B2AE3D27D4EB4Full; d = (d ^ w[i + 3]) * 0x165667B19E3779F9ull;
    }
    for (; i < n; ++i) a = (a ^ w[i]) * 0x100000001B3ull;
    const unsigned char *t = (const unsigned char *)data + 8 * n;
    for (size_t k = 0; k < bytes - 8 * n; ++k) b = (b ^ t[k]) * 0x100000001B3ull;
    return (a ^ (b << 1) ^ (c >> 1) ^ (d << 7)) * 0xFF51AFD7ED558CCDull;
}

DropinKey dropin_key(const CSR_Matrix *d, const CSR_Matrix *o, const INFO_Matrix *info, Comm *comm)
{
    DropinKey k{};
    k.dv = d->val; k.dc = d->col; k.dp = d->ptr; k.ov = o->val; k.oc = o->col; k.op = o->ptr;
    k.rows = d->rows; k.nnz_d = d->rows ? d->ptr[d->rows] : 0u; k.n_glob = info->rows;
    k.nranks = comm->nranks; k.rank = comm->rank; k.comm = comm; k.p2p = comm->p2p;
    k.nnz_o = (comm->nranks > 1 && o->rows) ? o->ptr[o->rows] : 0u;
    k.first_row = info->displs ? info->displs[comm->rank] : 0;
    uint64_t h = 0x243F6A8885A308D3ull;
    h = hash_words(h, d->ptr, sizeof(unsigned) * ((size_t)d->rows + 1));
    h = hash_words(h, d->col, sizeof(unsigned) * (size_t)k.nnz_d);
    h = hash_words(h, d->val, sizeof(double) * (size_t)k.nnz_d);
    if (comm->nranks > 1) {
        h = hash_words(h, o->ptr, sizeof(unsigned) * ((size_t)o->rows + 1));
        h = hash_words(h, o->col, sizeof(unsigned) * (size_t)k.nnz_o);
        h = hash_words(h, o->val, sizeof(double) * (size_t)k.nnz_o);
        h = hash_words(h, info->recvcounts, sizeof(int) * (size_t)comm->nranks);
        h = hash_words(h, info->displs, sizeof(int) * (size_t)comm->nranks);
    }
    k.hash = h;
    return k;
}

// every rank contributes one flag; true when it is set on all of them

bool dropin_cache_enabled()
{
    static const bool enabled = !(getenv("BICG_DROPIN_CACHE") && atoi(getenv("BICG_DROPIN_CACHE")) == 0);
    return enabled;
}

// the resident context for these blocks: reused when nothing changed, rebuilt otherwise (collective)
bicg_ctx *dropin_context(const CSR_Matrix *diag, const CSR_Matrix *offd, const INFO_Matrix *info)
{
    Comm *comm = comm_get();
    if (!dropin_cache_enabled()) {
        // create / destroy per call, but the library keeps ownership all the same (a caller that asked for the context
        // through bicg_dropin_context must not be left with one to free, and must not meet a SECOND copy of the matrix
        // on the GPU when it calls a solver next): the previous context goes before the new one is built
        if (g_dropin.ctx) { bicg_destroy(g_dropin.ctx); g_dropin.ctx = nullptr; }
        g_dropin.misses++;
        g_dropin.ctx = bicg_create(diag, offd, info);
        g_dropin.key = DropinKey{};
        return g_dropin.ctx;
    }
    const DropinKey key = dropin_key(diag, offd, info, comm);
    const bool hit = all_ranks(comm, g_dropin.ctx != nullptr && g_dropin.key == key);
    if (hit) { g_dropin.hits++; return g_dropin.ctx; }
    if (g_dropin.ctx) { bicg_destroy(g_dropin.ctx); g_dropin.ctx = nullptr; }
    g_dropin.misses++;
    g_dropin.ctx = bicg_create(diag, offd, info);
    g_dropin.key = key;
    return g_dropin.ctx;
}
void dropin_release(bicg_ctx *c)
{
    // caching disabled: the context does not outlive the solver call
    if (c && !dropin_cache_enabled() && c == g_dropin.ctx) { bicg_destroy(c); g_dropin.ctx = nullptr; }
}

// drop-in fallback from an automatically chosen peer-to-peer path. p2p_guard: arm it for this call (keeps copies of the
// caller's vectors); p2p_fell_back: collective -- true when some rank timed out; the resident context and the
// peer-to-peer state are gone then, and the next dropin_context() builds on the transport's collectives.
bool p2p_guard(bicg_ctx *c, const double *x, const double *r, std::vector<double> &x0, std::vector<double> &b)
{
    if (!c->p2p || !c->comm->p2p_auto) return false;
    c->soft_fail = true;
    const size_t nuser = c->phantom ? 0 : c->n_loc;      // a rank without rows: the caller's vectors are empty
    x0.assign(x, x + nuser); b.assign(r, r + nuser);
    return true;
}
bool p2p_fell_back(bicg_ctx *c)
{
    Comm *comm = c->comm;
    const bool failed = !all_ranks(comm, !c->comm_failed);
    if (!failed) return false;
    if (comm->rank == 0)
        fprintf(stderr, "bicgstab_hip: the peer-to-peer data path timed out in a solve although its self-test had passed; "
                        "repeating the solve with the %s collectives\n", comm->name());
    bicg_dropin_release();
    if (!g_live.empty()) die("peer-to-peer transport", "timed out, and other contexts still use it: cannot fall back");
    p2p_disable(comm);
    return true;
}

int dropin(int method, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info, double *x, double *r, int krr, int nrr)
{
    check_square(info);
    bicg_options o;
    env_options(&o);
    o.krr = krr; o.nrr = nrr;
    bicg_ctx *c = dropin_context(diag, offd, info);
    if (!c) die("bicg_create", "failed");
    bicg_result res;
    // The peer-to-peer data path is chosen automatically when its self-test passes (bicg_comm_init_mpi "auto"). Should it
    // fail in a real solve all the same -- a wait for a peer times out -- the solve is repeated on the transport's own
    // collectives (RCCL / MPI-staged) from the caller's x0 and b instead of ending the program.
    std::vector<double> x0, b;
    const bool guarded = p2p_guard(c, x, r, x0, b);
    int k = bicg_solve(c, method, x, r, &o, &res);
    if (guarded && p2p_fell_back(c)) {
        memcpy(x, x0.data(), sizeof(double) * x0.size());
        memcpy(r, b.data(), sizeof(double) * b.size());
        c = dropin_context(diag, offd, info);
        if (!c) die("bicg_create", "failed");
        k = bicg_solve(c, method, x, r, &o, &res);
    }
    dropin_release(c);
    return k;
}


extern "C" {

bicg_ctx *bicg_dropin_context(const CSR_Matrix *diag, const CSR_Matrix *offd, const INFO_Matrix *info)
{
    return dropin_context(diag, offd, info);
}
void bicg_dropin_release(void)
{
    if (g_dropin.ctx) { bicg_destroy(g_dropin.ctx); g_dropin.ctx = nullptr; }
}

void bicg_dropin_stats(unsigned int *hits, unsigned int *misses)
{
    if (hits) *hits = g_dropin.hits;
    if (misses) *misses = g_dropin.misses;
}


// BICG_DISPLAY_ERROR=1: what the reference prints when it is compiled with -DDISPLAY_ERROR (src/shifted_switching_solver.c:327-335,
// 570-598): the right-hand side is formed once more as ans = (A + sigma[seed] I) 1 -- what its drivers pass as b, src/main_shifted.c
// -- and every system's || (A + sigma_i I) x_i - ans || / || ans || is printed for the seed ("0, ...") and every tenth shift
// ("1, ..."). Here: one product on the device for ans, then the batched residuals of bicg_shifted_residuals (the matrix read once
// per 16 shifts). Collective like the solve itself.
static void display_error(bicg_ctx *c, const double *x_set, const double *sigma, int nsig, int seed)
{
    const size_t n = c->phantom ? 0 : c->n_loc;
    std::vector<double> ones(std::max<size_t>(n, 1), 1.0), ans(std::max<size_t>(n, 1), 0.0), err((size_t)nsig, 0.0);
    bicg_spmv(c, ones.data(), ans.data());
    for (size_t j = 0; j < n; ++j) ans[j] += sigma[seed] * ones[j];                        // my_daxpy(sigma[seed], temp, ans_loc)
    bicg_shifted_residuals(c, x_set, ans.data(), sigma, nsig, err.data());
    if (c->rank != 0) return;
    printf("seed(0:seed, 1:shift), sigma, relative error\n");
    for (int i = 0; i < nsig; ++i) {
        if (i == seed) printf("0, %e, %e\n", sigma[i], err[i]);
        else if (i % 10 == 0) printf("1, %e, %e\n", sigma[i], err[i]);
    }
    fflush(stdout);
}

static int dropin_shifted(int mode, CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x_set, double *r, double *sigma, int nsig, int seed)
{
    check_square(i);
    bicg_options opt;
    env_options(&opt);
    if (!getenv("BICG_TOL")) opt.tol = 1.0e-12;      // EPS of reference src/shifted_solver.c:5
    bicg_ctx *c = dropin_context(d, o, i);
    if (!c) die("bicg_create", "failed");
    bicg_result res;
    std::vector<double> x0, b, xs0;
    const bool guarded = p2p_guard(c, x_set, r, x0, b);
    if (guarded) xs0.assign(x_set, x_set + (size_t)nsig * c->n_loc);
    int k = run_shifted(c, mode, x_set, r, sigma, nsig, seed, &opt, &res);
    if (guarded && p2p_fell_back(c)) {
        memcpy(x_set, xs0.data(), sizeof(double) * xs0.size());
        memcpy(r, b.data(), sizeof(double) * b.size());
        c = dropin_context(d, o, i);
        if (!c) die("bicg_create", "failed");
        k = run_shifted(c, mode, x_set, r, sigma, nsig, seed, &opt, &res);
    }
    if (getenv("BICG_DISPLAY_ERROR") && atoi(getenv("BICG_DISPLAY_ERROR")) != 0) display_error(c, x_set, sigma, nsig, mode == SH_XI ? 0 : seed);
    dropin_release(c);
    return k;
}

// ---- shifted drop-ins: reference src/shifted_solver.h:17-19. The three reference functions perform
// the same arithmetic in a different order (their outputs are bit-identical to each other).
int shifted_lopbicgstab(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n, int seed) { return dropin_shifted(SH_LOP, d, o, i, x, r, sigma, n, seed); }
int shifted_lopbicgstab_v2(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n, int seed) { return dropin_shifted(SH_LOP, d, o, i, x, r, sigma, n, seed); }
int shifted_lopbicgstab_nooverlap(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n, int seed) { return dropin_shifted(SH_LOP, d, o, i, x, r, sigma, n, seed); }
// src/shifted_solver.h:20-21 (the two reference functions are bit-identical to each other)
int shifted_pipe_lopbicgstab(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n, int seed) { return dropin_shifted(SH_PIPE, d, o, i, x, r, sigma, n, seed); }
int shifted_pipe_lopbicgstab_nooverlap(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n, int seed) { return dropin_shifted(SH_PIPE, d, o, i, x, r, sigma, n, seed); }
// src/shifted_solver.h:16 (seed system = A, shift index 0)
int shifted_bicgstab(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n) { return dropin_shifted(SH_XI, d, o, i, x, r, sigma, n, 0); }
// reference src/shifted_switching_solver.h:10-12
int shifted_lopbicg(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n, int seed) { return dropin_shifted(SH_FLAG, d, o, i, x, r, sigma, n, seed); }
int shifted_lopbicg_switching(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n, int seed) { return dropin_shifted(SH_SWITCH, d, o, i, x, r, sigma, n, seed); }
int shifted_lopbicg_switching_noovlp(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, double *sigma, int n, int seed) { return dropin_shifted(SH_SWITCH, d, o, i, x, r, sigma, n, seed); }

// ---- drop-in entry points: reference src/solver.h:10-13
int bicgstab(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r) { return dropin(BICG_BICGSTAB, d, o, i, x, r, 0, 0); }
int ca_bicgstab(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r) { return dropin(BICG_CA_BICGSTAB, d, o, i, x, r, 0, 0); }
int pipe_bicgstab(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r) { return dropin(BICG_PIPE_BICGSTAB, d, o, i, x, r, 0, 0); }
int pipe_bicgstab_rr(CSR_Matrix *d, CSR_Matrix *o, INFO_Matrix *i, double *x, double *r, int krr, int nrr)
{
    return dropin(BICG_PIPE_BICGSTAB_RR, d, o, i, x, r, krr, nrr);
}

}  // extern "C"
