/*
 * bicg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See bicg_oracle.h.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no FMA contraction, so every a*b+c is two
 * roundings exactly as in an -O2 x86-64 build of the reference without -march=native).
 * Every function names the reference lines it restates.
 */
#include "bicg_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ partition */

/* reference src/matrix.c:295-308: m/P rows each, the first m%P ranks get one more. */
void orc_partition(unsigned n, int P, int *counts, int *displs)
{
    int base = (int)(n / (unsigned)P), extra = (int)(n % (unsigned)P);
    for (int p = 0; p < P; ++p) {
        int lo = p * base + (p < extra ? p : extra);
        counts[p] = base + (p < extra ? 1 : 0);
        displs[p] = lo;
    }
}

static void csr_alloc(orc_csr *A, unsigned rows, unsigned cols, unsigned nz)
{
    A->rows = rows; A->cols = cols; A->nz = nz;
    A->val = (double *)calloc(nz ? nz : 1, sizeof(double));
    A->col = (unsigned *)calloc(nz ? nz : 1, sizeof(unsigned));
    A->ptr = (unsigned *)calloc((size_t)rows + 1, sizeof(unsigned));
}

static void csr_release(orc_csr *A) { free(A->val); free(A->col); free(A->ptr); }

/*
 * reference src/matrix.c:336-340 and 380-392 classify each triplet of the file as diag (row and
 * column inside the rank's range, column made local) or offd (row inside, column kept GLOBAL);
 * coo2csr (206-232) then orders by row with a stable merge sort, i.e. file order inside a row.
 * A counting sort on the row key that scans the triplets in file order produces the same CSR.
 */
orc_dist *orc_dist_from_coo(unsigned n, unsigned nnz, const unsigned *row, const unsigned *col,
                            const double *val, int P)
{
    return orc_dist_from_coo_part(n, nnz, row, col, val, P, NULL);
}

/* Same with a caller-supplied contiguous partition (rows per rank); counts == NULL = the reference's
 * equal-rows partition. The reference's live code has no other partition (its nnz-balanced
 * DYNAMIC_ROWS branch sits unused in archive/matrix.c:407-446), so this entry is an extension: the
 * per-rank arithmetic is the same mult()/ddot restatement, only the cuts move. */
orc_dist *orc_dist_from_coo_part(unsigned n, unsigned nnz, const unsigned *row, const unsigned *col,
                                 const double *val, int P, const int *counts)
{
    orc_dist *d = (orc_dist *)calloc(1, sizeof(orc_dist));
    d->P = P; d->n = n;
    d->counts = (int *)malloc(sizeof(int) * (size_t)P);
    d->displs = (int *)malloc(sizeof(int) * (size_t)P);
    d->diag = (orc_csr *)calloc((size_t)P, sizeof(orc_csr));
    d->offd = (orc_csr *)calloc((size_t)P, sizeof(orc_csr));
    if (counts) {
        int at = 0;
        for (int p = 0; p < P; ++p) { d->counts[p] = counts[p]; d->displs[p] = at; at += counts[p]; }
    } else {
        orc_partition(n, P, d->counts, d->displs);
    }

    /* owner of every row */
    int *owner = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    for (int p = 0; p < P; ++p)
        for (int i = 0; i < d->counts[p]; ++i) owner[d->displs[p] + i] = p;

    /* pass 1: per-row counts for both blocks (global row index) */
    unsigned *cnt_d = (unsigned *)calloc((size_t)n + 1, sizeof(unsigned));
    unsigned *cnt_o = (unsigned *)calloc((size_t)n + 1, sizeof(unsigned));
    for (unsigned e = 0; e < nnz; ++e) {
        int p = owner[row[e]];
        unsigned lo = (unsigned)d->displs[p], hi = lo + (unsigned)d->counts[p];
        if (col[e] >= lo && col[e] < hi) cnt_d[row[e]]++; else cnt_o[row[e]]++;
    }
    for (int p = 0; p < P; ++p) {
        unsigned lo = (unsigned)d->displs[p], rows = (unsigned)d->counts[p];
        unsigned nd = 0, no = 0;
        for (unsigned i = 0; i < rows; ++i) { nd += cnt_d[lo + i]; no += cnt_o[lo + i]; }
        csr_alloc(&d->diag[p], rows, rows, nd);  /* src/matrix.c:343-345: cols = local rows */
        csr_alloc(&d->offd[p], rows, n, no);     /* src/matrix.c:350-352: cols = n (global) */
        unsigned ad = 0, ao = 0;
        for (unsigned i = 0; i < rows; ++i) {
            d->diag[p].ptr[i] = ad; ad += cnt_d[lo + i];
            d->offd[p].ptr[i] = ao; ao += cnt_o[lo + i];
        }
        d->diag[p].ptr[rows] = ad; d->offd[p].ptr[rows] = ao;
    }
    /* pass 2: scatter in file order (fill cursors reuse cnt arrays) */
    memset(cnt_d, 0, sizeof(unsigned) * ((size_t)n + 1));
    memset(cnt_o, 0, sizeof(unsigned) * ((size_t)n + 1));
    for (unsigned e = 0; e < nnz; ++e) {
        int p = owner[row[e]];
        unsigned lo = (unsigned)d->displs[p], hi = lo + (unsigned)d->counts[p];
        unsigned li = row[e] - lo;
        if (col[e] >= lo && col[e] < hi) {
            unsigned k = d->diag[p].ptr[li] + cnt_d[row[e]]++;
            d->diag[p].val[k] = val[e]; d->diag[p].col[k] = col[e] - lo;
        } else {
            unsigned k = d->offd[p].ptr[li] + cnt_o[row[e]]++;
            d->offd[p].val[k] = val[e]; d->offd[p].col[k] = col[e];
        }
    }
    free(cnt_d); free(cnt_o); free(owner);
    return d;
}

void orc_dist_free(orc_dist *d)
{
    if (!d) return;
    for (int p = 0; p < d->P; ++p) { csr_release(&d->diag[p]); csr_release(&d->offd[p]); }
    free(d->diag); free(d->offd); free(d->counts); free(d->displs); free(d);
}

/*
 * Matrix-Market reader for the only flavour the reference's block loader handles correctly
 * ("coordinate real general"; src/matrix.c:363-378 never assigns val for pattern/integer and
 * never mirrors symmetric storage -- SURVEY.md section 4 defect 2). 1-based -> 0-based as in
 * src/matrix.c:333-334.
 */
int orc_read_mtx(const char *path, unsigned *n_rows, unsigned *n_cols, unsigned *nnz,
                 unsigned **row, unsigned **col, double **val)
{
    FILE *f = fopen(path, "r");
    if (!f) return 1;
    char line[1100];
    if (!fgets(line, sizeof line, f) || strncmp(line, "%%MatrixMarket", 14) != 0) { fclose(f); return 2; }
    if (!strstr(line, "coordinate") || !strstr(line, "real") || !strstr(line, "general")) { fclose(f); return 3; }
    do { if (!fgets(line, sizeof line, f)) { fclose(f); return 4; } } while (line[0] == '%');
    unsigned m, n, nz;
    if (sscanf(line, "%u %u %u", &m, &n, &nz) != 3) { fclose(f); return 5; }
    *row = (unsigned *)malloc(sizeof(unsigned) * (size_t)(nz ? nz : 1));
    *col = (unsigned *)malloc(sizeof(unsigned) * (size_t)(nz ? nz : 1));
    *val = (double *)malloc(sizeof(double) * (size_t)(nz ? nz : 1));
    for (unsigned e = 0; e < nz; ++e) {
        unsigned i, j; double v;
        if (fscanf(f, "%u %u %lg", &i, &j, &v) != 3) { fclose(f); return 6; }
        (*row)[e] = i - 1; (*col)[e] = j - 1; (*val)[e] = v;
    }
    fclose(f);
    *n_rows = m; *n_cols = n; *nnz = nz;
    return 0;
}

/* ------------------------------------------------------------------ kernels */

/* reference src/matrix.c:498-516: per row a fresh accumulator summed in stored order, then
 * ADDED to y (y is not overwritten). */
void orc_mult(const orc_csr *A, const double *x, double *y)
{
    for (unsigned i = 0; i < A->rows; ++i) {
        double acc = 0.0;
        for (unsigned j = A->ptr[i]; j < A->ptr[i + 1]; ++j) acc += A->val[j] * x[A->col[j]];
        y[i] += acc;
    }
}

/* reference src/matrix.c:428-441: allgather x (here: x already is the concatenation), zero y,
 * diag product with the rank's own slice of x, then offd product with the full x. */
void orc_spmv(const orc_dist *d, const double *x, double *y)
{
    for (int p = 0; p < d->P; ++p) {
        double *yp = y + d->displs[p];
        for (int i = 0; i < d->counts[p]; ++i) yp[i] = 0.0;
        orc_mult(&d->diag[p], x + d->displs[p], yp);
        orc_mult(&d->offd[p], x, yp);
    }
}

/* reference src/vector.c:3-7 */
void orc_daxpy(int n, double a, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] += a * x[i]; }
/* reference src/vector.c:9-15 */
double orc_ddot(int n, const double *x, const double *y) { double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * y[i]; return s; }
/* reference src/vector.c:17-21 */
void orc_dscal(int n, double a, double *x) { for (int i = 0; i < n; ++i) x[i] *= a; }
/* reference src/vector.c:23-27 */
void orc_dcopy(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = x[i]; }

/* my_ddot per rank + MPI_Iallreduce(MPI_SUM) (e.g. src/solver.c:89-91). MPI leaves the
 * association of the P partial sums unspecified; MPICH's recursive doubling gives the balanced
 * pairwise tree ((p0+p1)+(p2+p3))+... for power-of-two P, which is what is restated here
 * (bit-identical to the reference under conda MPICH 3.3.2 at P = 1, 2, 4, 8 -- see
 * tests/test_oracle_golden.py). Other P: pairs first, leftovers carried upward. */
double orc_dist_dot(const orc_dist *d, const double *x, const double *y)
{
    double part[1024];
    int m = d->P;
    if (m > 1024) m = 1024;
    for (int p = 0; p < m; ++p)
        part[p] = orc_ddot(d->counts[p], x + d->displs[p], y + d->displs[p]);
    while (m > 1) {
        int h = 0;
        for (int p = 0; p + 1 < m; p += 2) part[h++] = part[p] + part[p + 1];
        if (m & 1) part[h++] = part[m - 1];
        m = h;
    }
    return part[0];
}

/* ------------------------------------------------------------------ solvers */

static double *vec_new(unsigned n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

static void trace_put(orc_opts *o, int k, double a, double w, double b, double rr)
{
    if (k < 1 || k > o->max_iter) return;
    if (o->tr_alpha) o->tr_alpha[k - 1] = a;
    if (o->tr_omega) o->tr_omega[k - 1] = w;
    if (o->tr_beta)  o->tr_beta[k - 1]  = b;
    if (o->tr_dotr)  o->tr_dotr[k - 1]  = rr;
}

/* u <- add + beta (u - omega v): the 3-call pattern daxpy(-omega) / dscal(beta) / daxpy(1.0)
 * of src/solver.c:217-219, 220-222, 352-360. */
static void recur3(int n, double omega, double beta, const double *v, const double *add, double *u)
{
    orc_daxpy(n, -omega, v, u);
    orc_dscal(n, beta, u);
    orc_daxpy(n, 1.0, add, u);
}

/* reference src/solver.c:35-146 */
static int solve_plain(const orc_dist *d, double *x, double *r, orc_opts *o)
{
    int n = (int)d->n, k = 0;
    double *Ax = vec_new(d->n), *rh = vec_new(d->n), *s = vec_new(d->n), *y = vec_new(d->n), *p = vec_new(d->n);
    double rTr, rTs, rTy, yTy, rTr_old, alpha = 0, beta = 0, omega = 0, dot_r, dot_zero;

    orc_spmv(d, x, Ax);                 /* :74 */
    orc_daxpy(n, -1.0, Ax, r);          /* :75  r = b - A x0 */
    orc_dcopy(n, r, rh);                /* :76 */
    orc_dcopy(n, r, p);                 /* :77 */
    rTr = orc_dist_dot(d, r, r);        /* :78-80 */
    dot_r = rTr; dot_zero = rTr;        /* :82-83 */

    while (dot_r > o->tol * o->tol * dot_zero && k < o->max_iter) {     /* :86 */
        orc_spmv(d, p, s);                          /* :88 */
        rTs = orc_dist_dot(d, rh, s);               /* :89-91 */
        alpha = rTr / rTs;                          /* :93 */
        orc_daxpy(n, -alpha, s, r);                 /* :94  q (kept in r) */
        orc_spmv(d, r, y);                          /* :96 */
        rTy = orc_dist_dot(d, r, y);                /* :97 */
        yTy = orc_dist_dot(d, y, y);                /* :99 */
        omega = rTy / yTy;                          /* :104 */
        orc_daxpy(n, alpha, p, x);                  /* :105 */
        orc_daxpy(n, omega, r, x);                  /* :106 */
        orc_daxpy(n, -omega, y, r);                 /* :107 */
        dot_r = orc_dist_dot(d, r, r);              /* :108 */
        rTr_old = rTr;                              /* :110 */
        rTr = orc_dist_dot(d, rh, r);               /* :111 */
        beta = (alpha / omega) * (rTr / rTr_old);   /* :116 */
        orc_dscal(n, beta, p);                      /* :117 */
        orc_daxpy(n, 1.0, r, p);                    /* :118 */
        orc_daxpy(n, -beta * omega, s, p);          /* :119 */
        k++;
        trace_put(o, k, alpha, omega, beta, dot_r);
    }
    o->dot_r = dot_r; o->dot_zero = dot_zero;
    free(Ax); free(rh); free(s); free(y); free(p);
    return k;
}

/* reference src/solver.c:160-278. p, s, z and omega are read before they are written there
 * (SURVEY.md section 4 defect 1); they are DEFINED as zero here, which is what the reference
 * computes whenever its malloc'ed pages are fresh. */
static int solve_ca(const orc_dist *d, double *x, double *r, orc_opts *o)
{
    int n = (int)d->n, k = 0;
    double *Ax = vec_new(d->n), *rh = vec_new(d->n), *s = vec_new(d->n), *z = vec_new(d->n),
           *w = vec_new(d->n), *p = vec_new(d->n);
    double rTr, rTw, wTw, rTs, rTz, rTr_old, alpha, beta, omega = 0.0, dot_r, dot_zero;

    orc_spmv(d, x, Ax);                 /* :200 */
    orc_daxpy(n, -1.0, Ax, r);          /* :201 */
    orc_dcopy(n, r, rh);                /* :202 */
    rTr = orc_dist_dot(d, r, r);        /* :203 */
    orc_spmv(d, r, w);                  /* :205 */
    rTw = orc_dist_dot(d, r, w);        /* :206 */
    alpha = rTr / rTw;                  /* :210 */
    beta = 0;                           /* :211 */
    dot_r = rTr; dot_zero = rTr;

    while (dot_r > o->tol * o->tol * dot_zero && k < o->max_iter) {     /* :216 */
        recur3(n, omega, beta, s, r, p);            /* :217-219 */
        recur3(n, omega, beta, z, w, s);            /* :220-222 */
        orc_spmv(d, s, z);                          /* :224 */
        orc_daxpy(n, -alpha, s, r);                 /* :225 q */
        orc_daxpy(n, -alpha, z, w);                 /* :226 y */
        rTw = orc_dist_dot(d, r, w);                /* :227 (q,y) */
        wTw = orc_dist_dot(d, w, w);                /* :228 (y,y) */
        omega = rTw / wTw;                          /* :232 */
        orc_daxpy(n, alpha, p, x);                  /* :233 */
        orc_daxpy(n, omega, r, x);                  /* :234 */
        orc_daxpy(n, -omega, w, r);                 /* :235 */
        dot_r = orc_dist_dot(d, r, r);              /* :236 */
        orc_spmv(d, r, w);                          /* :238 */
        rTr_old = rTr;
        rTr = orc_dist_dot(d, rh, r);               /* :240 */
        rTw = orc_dist_dot(d, rh, w);               /* :241 */
        rTs = orc_dist_dot(d, rh, s);               /* :242 */
        rTz = orc_dist_dot(d, rh, z);               /* :243 */
        double alpha_used = alpha;
        beta = (alpha / omega) * (rTr / rTr_old);               /* :248 */
        alpha = rTr / (rTw + beta * (rTs - omega * rTz));       /* :249 */
        k++;
        trace_put(o, k, alpha_used, omega, beta, dot_r);
    }
    o->dot_r = dot_r; o->dot_zero = dot_zero;
    free(Ax); free(rh); free(s); free(z); free(w); free(p);
    return k;
}

/* reference src/solver.c:292-417 (rr == 0) and 433-576 (rr != 0: residual replacement when
 * k % krr == 0 && k > 0 && k <= krr*nrr, :498 and :522). Same zero definition of the
 * uninitialised p, s, z, v, omega as solve_ca. */
static int solve_pipe(const orc_dist *d, double *x, double *r, orc_opts *o, int rr)
{
    int n = (int)d->n, k = 0;
    double *b = vec_new(d->n), *Ax = vec_new(d->n), *rh = vec_new(d->n), *s = vec_new(d->n),
           *z = vec_new(d->n), *w = vec_new(d->n), *p = vec_new(d->n), *v = vec_new(d->n),
           *t = vec_new(d->n);
    double rTr, rTw, wTw, rTs, rTz, rTr_old, alpha, beta, omega = 0.0, dot_r, dot_zero;

    if (rr) orc_dcopy(n, r, b);         /* :475 */
    orc_spmv(d, x, Ax);                 /* :333 */
    orc_daxpy(n, -1.0, Ax, r);          /* :334 */
    orc_dcopy(n, r, rh);                /* :335 */
    rTr = orc_dist_dot(d, r, r);        /* :336 */
    orc_spmv(d, r, w);                  /* :338 */
    rTw = orc_dist_dot(d, r, w);        /* :339 */
    orc_spmv(d, w, t);                  /* :341 */
    alpha = rTr / rTw;                  /* :345 */
    beta = 0;                           /* :346 */
    dot_r = rTr; dot_zero = rTr;

    while (dot_r > o->tol * o->tol * dot_zero && k < o->max_iter) {     /* :351 */
        int replace = rr && (k % o->krr == 0) && k > 0 && k <= o->krr * o->nrr;
        recur3(n, omega, beta, s, r, p);            /* :352-354 */
        if (replace) {
            orc_spmv(d, p, s);                      /* :499 */
            orc_spmv(d, s, z);                      /* :500 */
        } else {
            recur3(n, omega, beta, z, w, s);        /* :355-357 */
            recur3(n, omega, beta, v, t, z);        /* :358-360 */
        }
        orc_daxpy(n, -alpha, s, r);                 /* :361 q */
        orc_daxpy(n, -alpha, z, w);                 /* :362 y */
        rTw = orc_dist_dot(d, r, w);                /* :363 */
        wTw = orc_dist_dot(d, w, w);                /* :364 */
        orc_spmv(d, z, v);                          /* :365 */
        omega = rTw / wTw;                          /* :369 */
        orc_daxpy(n, alpha, p, x);                  /* :370 */
        orc_daxpy(n, omega, r, x);                  /* :371 */
        if (replace) {
            orc_spmv(d, x, Ax);                     /* :523 */
            orc_dcopy(n, b, r);                     /* :524 */
            orc_daxpy(n, -1.0, Ax, r);              /* :525 */
            orc_spmv(d, r, w);                      /* :526 */
        } else {
            orc_daxpy(n, -omega, w, r);             /* :372 */
            orc_daxpy(n, -alpha, v, t);             /* :374 */
            orc_daxpy(n, -omega, t, w);             /* :375 */
        }
        dot_r = orc_dist_dot(d, r, r);              /* :373 / :533 */
        rTr_old = rTr;
        rTr = orc_dist_dot(d, rh, r);               /* :377 */
        rTw = orc_dist_dot(d, rh, w);               /* :378 */
        rTs = orc_dist_dot(d, rh, s);               /* :379 */
        rTz = orc_dist_dot(d, rh, z);               /* :380 */
        orc_spmv(d, w, t);                          /* :381 */
        double alpha_used = alpha;
        beta = (alpha / omega) * (rTr / rTr_old);               /* :387 */
        alpha = rTr / (rTw + beta * (rTs - omega * rTz));       /* :388 */
        k++;
        trace_put(o, k, alpha_used, omega, beta, dot_r);
    }
    o->dot_r = dot_r; o->dot_zero = dot_zero;
    free(b); free(Ax); free(rh); free(s); free(z); free(w); free(p); free(v); free(t);
    return k;
}

/* y = (A + sigma I) x : MPI_csr_spmv_ovlap followed by my_daxpy(sigma, x, y), e.g. src/shifted_solver.c:261-262 */
static void spmv_shift(const orc_dist *d, double sigma, const double *x, double *y)
{
    orc_spmv(d, x, y);
    orc_daxpy((int)d->n, sigma, x, y);
}

/* reference src/shifted_solver.c:182-354 */
int orc_shifted_lop(const orc_dist *d, double *x_set, double *r, const double *sigma, int nsig, int seed, orc_opts *o)
{
    const int n = (int)d->n;
    int k = 0;
    double *r_old = vec_new(d->n), *rh = vec_new(d->n), *s = vec_new(d->n), *y = vec_new(d->n);
    double *p_set = (double *)calloc((size_t)n * (size_t)nsig + 1, sizeof(double));      /* :223 calloc */
    double *alpha = vec_new(nsig), *beta = vec_new(nsig), *omega = vec_new(nsig), *eta = vec_new(nsig),
           *zeta = vec_new(nsig), *pi_new = vec_new(nsig), *pi_old = vec_new(nsig);
    double alpha_old, beta_old, dot_r, dot_zero, rTr, rTs, qTq, qTy, rTr_old, max_zeta_pi;
#define P_(j) (p_set + (size_t)(j) * (size_t)n)
#define X_(j) (x_set + (size_t)(j) * (size_t)n)

    rTr = orc_dist_dot(d, r, r);                    /* :238 */
    orc_dcopy(n, r, rh);                            /* :240 */
    for (int i = 0; i < nsig; ++i) {                /* :241-249 */
        beta[i] = 0.0; alpha[i] = 1.0; eta[i] = 0.0; pi_old[i] = 1.0; pi_new[i] = 1.0; zeta[i] = 1.0;
    }
    orc_dcopy(n, r, P_(seed));                      /* :250 */
    dot_r = rTr; dot_zero = rTr; max_zeta_pi = 1.0; /* :253-255 */

    while (max_zeta_pi * max_zeta_pi * dot_r > o->tol * o->tol * dot_zero && k < o->max_iter) {   /* :257 */
        spmv_shift(d, sigma[seed], P_(seed), s);            /* :259-260 */
        rTs = orc_dist_dot(d, rh, s);                       /* :261 */
        for (int j = 0; j < nsig; ++j) {                    /* :262-267 */
            if (j == seed) continue;
            beta[j] = (pi_old[j] / pi_new[j]) * (pi_old[j] / pi_new[j]) * beta[seed];
            orc_dscal(n, beta[j], P_(j));
            orc_daxpy(n, 1.0 / (pi_new[j] * zeta[j]), r, P_(j));
        }
        orc_dcopy(nsig, pi_new, pi_old);                    /* :268 */
        orc_dcopy(n, r, r_old);                             /* :269 */
        alpha_old = alpha[seed]; beta_old = beta[seed];     /* :270-271 */

        alpha[seed] = rTr / rTs;                            /* :274 */
        orc_daxpy(n, -alpha[seed], s, r);                   /* :275  q */
        spmv_shift(d, sigma[seed], r, y);                   /* :276-277 */
        qTq = orc_dist_dot(d, r, r);                        /* :279 */
        qTy = orc_dist_dot(d, r, y);                        /* :280 */
        for (int j = 0; j < nsig; ++j) {                    /* :281-287 */
            if (j == seed) continue;
            eta[j] = (beta_old / alpha_old) * alpha[seed] * eta[j] - (sigma[seed] - sigma[j]) * alpha[seed] * pi_old[j];
            pi_new[j] = eta[j] + pi_old[j];
            alpha[j] = (pi_old[j] / pi_new[j]) * alpha[seed];
        }
        omega[seed] = qTq / qTy;                            /* :291 */
        orc_daxpy(n, alpha[seed], P_(seed), X_(seed));      /* :292 */
        orc_daxpy(n, omega[seed], r, X_(seed));             /* :293 */
        for (int j = 0; j < nsig; ++j) {                    /* :294-302 */
            if (j == seed) continue;
            omega[j] = omega[seed] / (1.0 - omega[seed] * (sigma[seed] - sigma[j]));
            orc_daxpy(n, omega[j] / (pi_new[j] * zeta[j]), r, X_(j));
            orc_daxpy(n, alpha[j], P_(j), X_(j));
            orc_daxpy(n, omega[j] / (alpha[j] * zeta[j] * pi_new[j]), r, P_(j));
            orc_daxpy(n, -omega[j] / (alpha[j] * zeta[j] * pi_old[j]), r_old, P_(j));
            zeta[j] = (1.0 - omega[seed] * (sigma[seed] - sigma[j])) * zeta[j];
        }
        orc_daxpy(n, -omega[seed], y, r);                   /* :303 */
        dot_r = orc_dist_dot(d, r, r);                      /* :304 */
        rTr_old = rTr;
        rTr = orc_dist_dot(d, rh, r);                       /* :306 */
        beta[seed] = (alpha[seed] / omega[seed]) * (rTr / rTr_old);   /* :310 */
        max_zeta_pi = 1.0;                                  /* :311-316 */
        for (int j = 0; j < nsig; ++j) {
            if (j == seed) continue;
            double a = 1.0 / (zeta[j] * pi_new[j]);
            if (a < 0) a = -a;
            if (a > max_zeta_pi) max_zeta_pi = a;
        }
        orc_dscal(n, beta[seed], P_(seed));                 /* :317 */
        orc_daxpy(n, 1.0, r, P_(seed));                     /* :318 */
        orc_daxpy(n, -beta[seed] * omega[seed], s, P_(seed));   /* :319 */
        k++;
        trace_put(o, k, alpha[seed], omega[seed], beta[seed], dot_r);
    }
#undef P_
#undef X_
    o->dot_r = dot_r; o->dot_zero = dot_zero;
    free(r_old); free(rh); free(s); free(y); free(p_set);
    free(alpha); free(beta); free(omega); free(eta); free(zeta); free(pi_new); free(pi_old);
    return k;
}

/* reference src/shifted_solver.c:703-895. omega[seed], s, z, v are read before they are written
 * there (:790-798, malloc'ed); defined as zero here like in the unshifted pipelined solver. */
int orc_shifted_pipe_lop(const orc_dist *d, double *x_set, double *r, const double *sigma, int nsig, int seed, orc_opts *o)
{
    const int n = (int)d->n;
    int k = 0;
    double *r_old = vec_new(d->n), *rh = vec_new(d->n), *s = vec_new(d->n), *z = vec_new(d->n), *w = vec_new(d->n),
           *v = vec_new(d->n), *t = vec_new(d->n);
    double *p_set = (double *)calloc((size_t)n * (size_t)nsig + 1, sizeof(double));
    double *alpha = vec_new(nsig), *beta = vec_new(nsig), *omega = vec_new(nsig), *eta = vec_new(nsig),
           *zeta = vec_new(nsig), *pi_new = vec_new(nsig), *pi_old = vec_new(nsig);
    double alpha_old, beta_old, dot_r, dot_zero, rTr, rTw, wTw, rTs, rTz, rTr_old, max_zeta_pi;
#define P_(j) (p_set + (size_t)(j) * (size_t)n)
#define X_(j) (x_set + (size_t)(j) * (size_t)n)
    const double sg = sigma[seed];

    rTr = orc_dist_dot(d, r, r);                            /* :762 */
    spmv_shift(d, sg, r, w);                                /* :764-765 */
    rTw = orc_dist_dot(d, r, w);                            /* :766 */
    spmv_shift(d, sg, w, t);                                /* :768-769 */
    orc_dcopy(n, r, rh);                                    /* :771 */
    for (int i = 0; i < nsig; ++i) { beta[i] = 0.0; alpha[i] = 1.0; eta[i] = 0.0; pi_old[i] = 1.0; pi_new[i] = 1.0; zeta[i] = 1.0; }
    orc_dcopy(n, r, P_(seed));                              /* :781 */
    alpha_old = 1.0;                                        /* :785 */
    alpha[seed] = rTr / rTw;                                /* :786 */
    dot_r = rTr; dot_zero = rTr; max_zeta_pi = 1.0;

    while (max_zeta_pi * max_zeta_pi * dot_r > o->tol * o->tol * dot_zero && k < o->max_iter) {   /* :792 */
        recur3(n, omega[seed], beta[seed], s, r, P_(seed));         /* :794-796 */
        recur3(n, omega[seed], beta[seed], z, w, s);                /* :797-799 */
        recur3(n, omega[seed], beta[seed], v, t, z);                /* :800-802 */
        for (int j = 0; j < nsig; ++j) {                            /* :803-808 */
            if (j == seed) continue;
            beta[j] = (pi_old[j] / pi_new[j]) * (pi_old[j] / pi_new[j]) * beta[seed];
            orc_dscal(n, beta[j], P_(j));
            orc_daxpy(n, 1.0 / (pi_new[j] * zeta[j]), r, P_(j));
        }
        orc_dcopy(n, r, r_old);                                     /* :809 */
        orc_daxpy(n, -alpha[seed], s, r);                           /* :810 q */
        orc_daxpy(n, -alpha[seed], z, w);                           /* :811 y */
        rTw = orc_dist_dot(d, r, w);                                /* :812 (q,y) */
        wTw = orc_dist_dot(d, w, w);                                /* :813 (y,y) */
        spmv_shift(d, sg, z, v);                                    /* :814-815 */
        orc_dcopy(nsig, pi_new, pi_old);                            /* :816 */
        beta_old = beta[seed];                                      /* :817 */
        for (int j = 0; j < nsig; ++j) {                            /* :818-824 */
            if (j == seed) continue;
            eta[j] = (beta_old / alpha_old) * alpha[seed] * eta[j] - (sg - sigma[j]) * alpha[seed] * pi_old[j];
            pi_new[j] = eta[j] + pi_old[j];
            alpha[j] = (pi_old[j] / pi_new[j]) * alpha[seed];
        }
        omega[seed] = rTw / wTw;                                    /* :828 */
        orc_daxpy(n, alpha[seed], P_(seed), X_(seed));              /* :829 */
        orc_daxpy(n, omega[seed], r, X_(seed));                     /* :830 */
        for (int j = 0; j < nsig; ++j) {                            /* :831-839 */
            if (j == seed) continue;
            omega[j] = omega[seed] / (1.0 - omega[seed] * (sg - sigma[j]));
            orc_daxpy(n, omega[j] / (pi_new[j] * zeta[j]), r, X_(j));
            orc_daxpy(n, alpha[j], P_(j), X_(j));
            orc_daxpy(n, omega[j] / (alpha[j] * zeta[j] * pi_new[j]), r, P_(j));
            orc_daxpy(n, -omega[j] / (alpha[j] * zeta[j] * pi_old[j]), r_old, P_(j));
            zeta[j] = (1.0 - omega[seed] * (sg - sigma[j])) * zeta[j];
        }
        orc_daxpy(n, -omega[seed], w, r);                           /* :840 */
        dot_r = orc_dist_dot(d, r, r);                              /* :841 */
        orc_daxpy(n, -alpha[seed], v, t);                           /* :842 */
        orc_daxpy(n, -omega[seed], t, w);                           /* :843 */
        rTr_old = rTr;
        rTr = orc_dist_dot(d, rh, r);                               /* :845 */
        rTw = orc_dist_dot(d, rh, w);                               /* :846 */
        rTs = orc_dist_dot(d, rh, s);                               /* :847 */
        rTz = orc_dist_dot(d, rh, z);                               /* :848 */
        spmv_shift(d, sg, w, t);                                    /* :849-850 */
        beta[seed] = (alpha[seed] / omega[seed]) * (rTr / rTr_old); /* :856 */
        alpha_old = alpha[seed];                                    /* :857 */
        alpha[seed] = rTr / (rTw + beta[seed] * (rTs - omega[seed] * rTz));   /* :858 */
        max_zeta_pi = 1.0;                                          /* :859-864 */
        for (int j = 0; j < nsig; ++j) {
            if (j == seed) continue;
            double a = 1.0 / (zeta[j] * pi_new[j]);
            if (a < 0) a = -a;
            if (a > max_zeta_pi) max_zeta_pi = a;
        }
        k++;
        trace_put(o, k, alpha_old, omega[seed], beta[seed], dot_r);
    }
#undef P_
#undef X_
    o->dot_r = dot_r; o->dot_zero = dot_zero;
    free(r_old); free(rh); free(s); free(z); free(w); free(v); free(t); free(p_set);
    free(alpha); free(beta); free(omega); free(eta); free(zeta); free(pi_new); free(pi_old);
    return k;
}

/* reference src/shifted_solver.c:13-180 */
int orc_shifted_bicgstab(const orc_dist *d, double *x_set, double *r, const double *sigma, int nsig, orc_opts *o)
{
    const int n = (int)d->n;
    int k = 0;
    double *r_old = vec_new(d->n), *rh = vec_new(d->n), *s = vec_new(d->n), *y = vec_new(d->n);
    double *p_set = (double *)calloc((size_t)n * (size_t)nsig + 1, sizeof(double));
    double *alpha = vec_new(nsig), *beta = vec_new(nsig), *omega = vec_new(nsig), *tau = vec_new(nsig),
           *xi_old = vec_new(nsig), *xi_curr = vec_new(nsig), *xi_new = vec_new(nsig);
    double alpha_old, beta_old, dot_r, dot_zero, rTr, rTs, rTy, yTy, rTr_old, max_xi;
#define P_(j) (p_set + (size_t)(j) * (size_t)n)
#define X_(j) (x_set + (size_t)(j) * (size_t)n)

    rTr = orc_dist_dot(d, r, r);                            /* :68 */
    orc_dcopy(n, r, rh);                                    /* :70 */
    for (int i = 0; i < nsig; ++i) {                        /* :71-78 */
        orc_dcopy(n, r, P_(i));
        beta[i] = 0.0; alpha[i] = 1.0; xi_old[i] = 1.0; xi_curr[i] = 1.0; tau[i] = 1.0;
    }
    dot_r = rTr; dot_zero = rTr; max_xi = 1.0;              /* :82-84 */

    while (max_xi * max_xi * dot_r > o->tol * o->tol * dot_zero && k < o->max_iter) {     /* :86 */
        orc_spmv(d, P_(0), s);                              /* :88 */
        rTs = orc_dist_dot(d, rh, s);                       /* :89 */
        for (int j = 1; j < nsig; ++j) {                    /* :90-94 */
            beta[j] = (xi_curr[j] / xi_old[j]) * (xi_curr[j] / xi_old[j]) * beta[0];
            orc_dscal(n, beta[j], P_(j));
            orc_daxpy(n, tau[j] * xi_curr[j], r, P_(j));
        }
        orc_dcopy(n, r, r_old);                             /* :95 */
        alpha_old = alpha[0]; beta_old = beta[0];           /* :96-97 */
        alpha[0] = rTr / rTs;                               /* :100 */
        orc_daxpy(n, -alpha[0], s, r);                      /* :102 */
        orc_spmv(d, r, y);                                  /* :103 */
        rTy = orc_dist_dot(d, r, y);                        /* :105 */
        yTy = orc_dist_dot(d, y, y);                        /* :106 */
        for (int j = 1; j < nsig; ++j) {                    /* :107-111 */
            xi_new[j] = (xi_curr[j] * xi_old[j] * alpha_old) /
                        (alpha[0] * beta_old * (xi_old[j] - xi_curr[j]) + xi_old[j] * alpha_old * (1.0 + alpha[0] * sigma[j]));
            alpha[j] = (xi_new[j] / xi_curr[j]) * alpha[0];
        }
        omega[0] = rTy / yTy;                               /* :115 */
        orc_daxpy(n, alpha[0], P_(0), X_(0));               /* :116 */
        orc_daxpy(n, omega[0], r, X_(0));                   /* :117 */
        for (int j = 1; j < nsig; ++j) {                    /* :118-124 */
            omega[j] = omega[0] / (1.0 + omega[0] * sigma[j]);
            orc_daxpy(n, omega[j] * tau[j] * xi_new[j], r, X_(j));
            orc_daxpy(n, alpha[j], P_(j), X_(j));
            orc_daxpy(n, omega[j] * tau[j] * xi_new[j] / alpha[j], r, P_(j));
            orc_daxpy(n, -omega[j] * tau[j] * xi_curr[j] / alpha[j], r_old, P_(j));
        }
        orc_daxpy(n, -omega[0], y, r);                      /* :125 */
        dot_r = orc_dist_dot(d, r, r);                      /* :126 */
        rTr_old = rTr;
        rTr = orc_dist_dot(d, rh, r);                       /* :128 */
        for (int j = 1; j < nsig; ++j) tau[j] = tau[j] / (1.0 + omega[0] * sigma[j]);     /* :129-131 */
        beta[0] = (alpha[0] / omega[0]) * (rTr / rTr_old);  /* :135 */
        max_xi = 1.0;                                       /* :136-140 */
        for (int j = 1; j < nsig; ++j) {
            double a = xi_curr[j] * tau[j];
            if (a < 0) a = -a;
            if (a > max_xi) max_xi = a;
        }
        orc_dcopy(nsig, xi_curr, xi_old);                   /* :141 */
        orc_dcopy(nsig, xi_new, xi_curr);                   /* :142 */
        orc_dscal(n, beta[0], P_(0));                       /* :143 */
        orc_daxpy(n, 1.0, r, P_(0));                        /* :144 */
        orc_daxpy(n, -beta[0] * omega[0], s, P_(0));        /* :145 */
        k++;
        trace_put(o, k, alpha[0], omega[0], beta[0], dot_r);
    }
#undef P_
#undef X_
    o->dot_r = dot_r; o->dot_zero = dot_zero;
    free(r_old); free(rh); free(s); free(y); free(p_set);
    free(alpha); free(beta); free(omega); free(tau); free(xi_old); free(xi_curr); free(xi_new);
    return k;
}

/* ------------------------------------------------------------------------------------------
 * Shifted solvers with per-shift convergence flags and SEED SWITCHING
 * (reference src/shifted_switching_solver.c).
 * ------------------------------------------------------------------------------------------ */

/* shifted_lopbicg, reference src/shifted_switching_solver.c:20-257: the "lop" recurrence in which a
 * shift whose residual bound |1/(zeta pi)| ||r|| has reached the tolerance is frozen (stop_flag);
 * the loop ends when every system has converged. stop_out[nsig] (optional) receives the flags. */
int orc_shifted_lopbicg(const orc_dist *d, double *x_set, double *r, const double *sigma, int nsig, int seed,
                        orc_opts *o, int *stop_out)
{
    const int n = (int)d->n;
    int k = 0, stop_count = 0;
    double *r_old = vec_new(d->n), *rh = vec_new(d->n), *s = vec_new(d->n), *y = vec_new(d->n);
    double *p_set = (double *)calloc((size_t)n * (size_t)nsig + 1, sizeof(double));      /* :59 */
    double *alpha = vec_new(nsig), *beta = vec_new(nsig), *omega = vec_new(nsig), *eta = vec_new(nsig),
           *zeta = vec_new(nsig), *pi_new = vec_new(nsig), *pi_old = vec_new(nsig);
    int *stop = (int *)calloc((size_t)nsig + 1, sizeof(int));                             /* :69 */
    double alpha_old, beta_old, dot_r, dot_zero, rTr, rTs, qTq, qTy, rTr_old;
#define P_(j) (p_set + (size_t)(j) * (size_t)n)
#define X_(j) (x_set + (size_t)(j) * (size_t)n)

    rTr = orc_dist_dot(d, r, r);                                /* :77 */
    orc_dcopy(n, r, rh);                                        /* :79 */
    for (int i = 0; i < nsig; ++i) {                            /* :80-88: EVERY p[sigma] starts as b */
        orc_dcopy(n, r, P_(i));
        alpha[i] = 1.0; beta[i] = 0.0; eta[i] = 0.0; pi_old[i] = 1.0; pi_new[i] = 1.0; zeta[i] = 1.0;
    }
    orc_dcopy(n, r, P_(seed));                                  /* :89 */
    dot_r = rTr; dot_zero = rTr;                                /* :92-93 */

    while (stop_count < nsig && k < o->max_iter) {              /* :100 */
        orc_dcopy(n, r, r_old);                                 /* :102 */
        orc_dcopy(nsig, pi_new, pi_old);                        /* :103 */
        alpha_old = alpha[seed]; beta_old = beta[seed];         /* :104-105 */
        spmv_shift(d, sigma[seed], P_(seed), s);                /* :107-108 */
        rTs = orc_dist_dot(d, rh, s);                           /* :110 */
        alpha[seed] = rTr / rTs;                                /* :113 */
        orc_daxpy(n, -alpha[seed], s, r);                       /* :114  q */
        spmv_shift(d, sigma[seed], r, y);                       /* :115-116 */
        qTq = orc_dist_dot(d, r, r);                            /* :117 */
        qTy = orc_dist_dot(d, r, y);                            /* :118 */
        omega[seed] = qTq / qTy;                                /* :122 */
        orc_daxpy(n, alpha[seed], P_(seed), X_(seed));          /* :123 */
        orc_daxpy(n, omega[seed], r, X_(seed));                 /* :124 */
        for (int j = 0; j < nsig; ++j) {                        /* :130-145 */
            if (j == seed || stop[j]) continue;
            eta[j] = (beta_old / alpha_old) * alpha[seed] * eta[j] - (sigma[seed] - sigma[j]) * alpha[seed] * pi_old[j];
            pi_new[j] = eta[j] + pi_old[j];
            alpha[j] = (pi_old[j] / pi_new[j]) * alpha[seed];
            omega[j] = omega[seed] / (1.0 - omega[seed] * (sigma[seed] - sigma[j]));
            orc_daxpy(n, omega[j] / (pi_new[j] * zeta[j]), r, X_(j));
            orc_daxpy(n, alpha[j], P_(j), X_(j));
            orc_daxpy(n, omega[j] / (alpha[j] * zeta[j] * pi_new[j]), r, P_(j));
            orc_daxpy(n, -omega[j] / (alpha[j] * zeta[j] * pi_old[j]), r_old, P_(j));
            zeta[j] = (1.0 - omega[seed] * (sigma[seed] - sigma[j])) * zeta[j];
        }
        orc_daxpy(n, -omega[seed], y, r);                       /* :152 */
        dot_r = orc_dist_dot(d, r, r);                          /* :153 */
        rTr_old = rTr;
        rTr = orc_dist_dot(d, rh, r);                           /* :155 */
        beta[seed] = (alpha[seed] / omega[seed]) * (rTr / rTr_old);     /* :159 */
        orc_dscal(n, beta[seed], P_(seed));                     /* :160 */
        orc_daxpy(n, 1.0, r, P_(seed));                         /* :161 */
        orc_daxpy(n, -beta[seed] * omega[seed], s, P_(seed));   /* :162 */
        for (int j = 0; j < nsig; ++j) {                        /* :164-170 */
            if (j == seed || stop[j]) continue;
            beta[j] = (pi_old[j] / pi_new[j]) * (pi_old[j] / pi_new[j]) * beta[seed];
            orc_dscal(n, beta[j], P_(j));
            orc_daxpy(n, 1.0 / (pi_new[j] * zeta[j]), r, P_(j));
        }
        for (int j = 0; j < nsig; ++j) {                        /* :180-199 */
            if (stop[j]) continue;
            const double a = j == seed ? 1.0 : fabs(1.0 / (zeta[j] * pi_new[j]));
            if (a * a * dot_r <= o->tol * o->tol * dot_zero) { stop[j] = 1; stop_count++; }
        }
        k++;                                                    /* :210 */
        trace_put(o, k, alpha[seed], omega[seed], beta[seed], dot_r);
    }
    if (stop_out) memcpy(stop_out, stop, sizeof(int) * (size_t)nsig);
    o->dot_r = dot_r; o->dot_zero = dot_zero;
    free(r_old); free(rh); free(s); free(y); free(p_set); free(stop);
    free(alpha); free(beta); free(omega); free(eta); free(zeta); free(pi_new); free(pi_old);
    return k;
}

/* shifted_lopbicg_switching (reference src/shifted_switching_solver.c:260-608) and its _noovlp twin
 * (:611-1016, same arithmetic, only the MPI_Wait placement and the section timers differ).
 * The seed system's alpha/beta/omega and every shift's pi are ARCHIVED per iteration; when the
 * seed converges first, the unconverged shift with the largest |1/(zeta pi)| becomes the new seed:
 * r is rescaled to its residual, the archives are rewritten for it and eta/pi/zeta of the
 * remaining shifts are re-derived from the rewritten history (:490-527).
 * k starts at 1 and the loop runs while k < max_iter + 1; the return value is k (iterations + 1).
 * Kept as written: (r#,r) is NOT rescaled together with r at a switch (:499, rTr keeps its value),
 * and `max_sigma` is read uninitialised when no unconverged shift has |1/(zeta pi)| > 1 (:462-467)
 * -- defined as 0 here, which is what the pinned reference build (zero-initialised locals) does.
 * final_seed / nswitch / stop_out are optional outputs for the tests. */
int orc_shifted_lopbicg_switching(const orc_dist *d, double *x_set, double *r, const double *sigma, int nsig, int seed,
                                  orc_opts *o, int *final_seed, int *nswitch, int *stop_out)
{
    const int n = (int)d->n;
    const int max_iter = o->max_iter + 1;                        /* :297 */
    int k = 1, stop_count = 0, max_sigma = 0, switches = 0;      /* :295 */
    double *r_old = vec_new(d->n), *rh = vec_new(d->n), *s = vec_new(d->n), *y = vec_new(d->n), *qc = vec_new(d->n);
    double *p_set = (double *)calloc((size_t)n * (size_t)nsig + 1, sizeof(double));
    double *alpha = vec_new(nsig), *beta = vec_new(nsig), *omega = vec_new(nsig), *eta = vec_new(nsig), *zeta = vec_new(nsig);
    double *a_arc = vec_new(max_iter + 1), *b_arc = vec_new(max_iter + 1), *w_arc = vec_new(max_iter + 1);
    double *pi = (double *)calloc((size_t)max_iter * (size_t)nsig + 1, sizeof(double));
    int *stop = (int *)calloc((size_t)nsig + 1, sizeof(int));
    double dot_r, dot_zero, rTr, rTs, qTq, qTy, rTr_old, max_zeta_pi;
#define PI_(j, i) pi[(size_t)(j) * (size_t)max_iter + (size_t)(i)]

    rTr = orc_dist_dot(d, r, r);                                /* :344 */
    orc_dcopy(n, r, rh);                                        /* :346 */
    for (int i = 0; i < nsig; ++i) {                            /* :347-355 */
        orc_dcopy(n, r, P_(i));
        alpha[i] = 1.0; beta[i] = 0.0; eta[i] = 0.0; PI_(i, 0) = 1.0; PI_(i, 1) = 1.0; zeta[i] = 1.0;
    }
    orc_dcopy(n, r, P_(seed));                                  /* :356 */
    dot_r = rTr; dot_zero = rTr;                                /* :359-360 */
    a_arc[0] = 1.0; b_arc[0] = 0.0;                             /* :363-364 */

    while (stop_count < nsig && k < max_iter) {                 /* :374 */
        orc_dcopy(n, r, r_old);                                 /* :376 */
        spmv_shift(d, sigma[seed], P_(seed), s);                /* :379-388 (spelled-out MPI_csr_spmv_ovlap) */
        rTs = orc_dist_dot(d, rh, s);                           /* :389 */
        a_arc[k] = rTr / rTs;                                   /* :392 */
        orc_daxpy(n, -a_arc[k], s, r);                          /* :393  q */
        orc_dcopy(n, r, qc);                                    /* :394  q_copy */
        spmv_shift(d, sigma[seed], r, y);                       /* :397-406 */
        qTq = orc_dist_dot(d, r, r);                            /* :407 */
        qTy = orc_dist_dot(d, r, y);                            /* :408 */
        w_arc[k] = qTq / qTy;                                   /* :412 */
        orc_daxpy(n, a_arc[k], P_(seed), X_(seed));             /* :413 */
        orc_daxpy(n, w_arc[k], r, X_(seed));                    /* :414 */
        orc_daxpy(n, -w_arc[k], y, r);                          /* :415 */
        dot_r = orc_dist_dot(d, r, r);                          /* :416 */
        rTr_old = rTr;
        rTr = orc_dist_dot(d, rh, r);                           /* :418 */
        b_arc[k] = (a_arc[k] / w_arc[k]) * (rTr / rTr_old);     /* :422 */
        orc_dscal(n, b_arc[k], P_(seed));                       /* :423 */
        orc_daxpy(n, 1.0, r, P_(seed));                         /* :424 */
        orc_daxpy(n, -b_arc[k] * w_arc[k], s, P_(seed));        /* :425 */

        for (int j = 0; j < nsig; ++j) {                        /* :431-446 */
            if (j == seed || stop[j]) continue;
            eta[j] = (b_arc[k - 1] / a_arc[k - 1]) * a_arc[k] * eta[j] - (sigma[seed] - sigma[j]) * a_arc[k] * PI_(j, k - 1);
            PI_(j, k) = eta[j] + PI_(j, k - 1);
            alpha[j] = (PI_(j, k - 1) / PI_(j, k)) * a_arc[k];
            omega[j] = w_arc[k] / (1.0 - w_arc[k] * (sigma[seed] - sigma[j]));
            orc_daxpy(n, omega[j] / (PI_(j, k) * zeta[j]), qc, X_(j));
            orc_daxpy(n, alpha[j], P_(j), X_(j));
            orc_daxpy(n, omega[j] / (alpha[j] * zeta[j] * PI_(j, k)), qc, P_(j));
            orc_daxpy(n, -omega[j] / (alpha[j] * zeta[j] * PI_(j, k - 1)), r_old, P_(j));
            zeta[j] = (1.0 - w_arc[k] * (sigma[seed] - sigma[j])) * zeta[j];
            beta[j] = (PI_(j, k - 1) / PI_(j, k)) * (PI_(j, k - 1) / PI_(j, k)) * b_arc[k];
            orc_dscal(n, beta[j], P_(j));
            orc_daxpy(n, 1.0 / (PI_(j, k) * zeta[j]), r, P_(j));
        }

        max_zeta_pi = 1.0;                                      /* :451-475 */
        for (int j = 0; j < nsig; ++j) {
            if (stop[j]) continue;
            const double a = j == seed ? 1.0 : fabs(1.0 / (zeta[j] * PI_(j, k)));
            if (a * a * dot_r <= o->tol * o->tol * dot_zero) {
                stop[j] = 1; stop_count++;
            } else if (a > max_zeta_pi) {
                max_zeta_pi = a; max_sigma = j;
            }
        }

        trace_put(o, k, a_arc[k], w_arc[k], b_arc[k], dot_r);   /* before a switch rewrites the archives */
        if (stop[seed] && stop_count < nsig) {                  /* :490-527 seed switching */
            const int ms = max_sigma;
            for (int i = 1; i <= k; ++i) {
                a_arc[i] = (PI_(ms, i - 1) / PI_(ms, i)) * a_arc[i];
                b_arc[i] = (PI_(ms, i - 1) / PI_(ms, i)) * (PI_(ms, i - 1) / PI_(ms, i)) * b_arc[i];
                w_arc[i] = w_arc[i] / (1.0 - w_arc[i] * (sigma[seed] - sigma[ms]));
            }
            orc_dscal(n, 1.0 / (zeta[ms] * PI_(ms, k)), r);     /* :499 */
            for (int j = 0; j < nsig; ++j) { eta[j] = 0.0; zeta[j] = 1.0; }
            for (int i = 1; i <= k; ++i)
                for (int j = 0; j < nsig; ++j) {
                    if (stop[j] || j == ms) continue;
                    eta[j] = (b_arc[i - 1] / a_arc[i - 1]) * a_arc[i] * eta[j] - (sigma[ms] - sigma[j]) * a_arc[i] * PI_(j, i - 1);
                    PI_(j, i) = eta[j] + PI_(j, i - 1);
                    zeta[j] = (1.0 - w_arc[i] * (sigma[ms] - sigma[j])) * zeta[j];
                }
            seed = ms;
            switches++;
        }
        k++;                                                    /* :536 */
    }
#undef PI_
#undef P_
#undef X_
    if (final_seed) *final_seed = seed;
    if (nswitch) *nswitch = switches;
    if (stop_out) memcpy(stop_out, stop, sizeof(int) * (size_t)nsig);
    o->dot_r = dot_r; o->dot_zero = dot_zero;
    free(r_old); free(rh); free(s); free(y); free(qc); free(p_set); free(stop); free(pi);
    free(alpha); free(beta); free(omega); free(eta); free(zeta); free(a_arc); free(b_arc); free(w_arc);
    return k;
}

/* which: 0 = shifted_lopbicg, 1 = shifted_lopbicg_switching (and _noovlp). info[0] = final seed,
 * info[1] = number of seed switches, info[2 .. 2+nsig) = stop flags. */
int orc_switching_coo(int which, int P, unsigned n, unsigned nnz, const unsigned *row, const unsigned *col,
                      const double *val, double *x_set, double *r, const double *sigma, int nsig, int seed, orc_opts *o,
                      int *info)
{
    orc_dist *d = orc_dist_from_coo(n, nnz, row, col, val, P);
    int k;
    info[0] = seed; info[1] = 0;
    if (which == 0) k = orc_shifted_lopbicg(d, x_set, r, sigma, nsig, seed, o, info + 2);
    else k = orc_shifted_lopbicg_switching(d, x_set, r, sigma, nsig, seed, o, &info[0], &info[1], info + 2);
    orc_dist_free(d);
    return k;
}

int orc_shifted_coo(int which, int P, unsigned n, unsigned nnz, const unsigned *row, const unsigned *col,
                    const double *val, double *x_set, double *r, const double *sigma, int nsig, int seed, orc_opts *o)
{
    orc_dist *d = orc_dist_from_coo(n, nnz, row, col, val, P);
    int k = which == 0 ? orc_shifted_lop(d, x_set, r, sigma, nsig, seed, o)
          : which == 1 ? orc_shifted_pipe_lop(d, x_set, r, sigma, nsig, seed, o)
                       : orc_shifted_bicgstab(d, x_set, r, sigma, nsig, o);
    orc_dist_free(d);
    return k;
}

int orc_shifted_lop_coo(int P, unsigned n, unsigned nnz, const unsigned *row, const unsigned *col,
                        const double *val, double *x_set, double *r, const double *sigma, int nsig, int seed,
                        orc_opts *o)
{
    orc_dist *d = orc_dist_from_coo(n, nnz, row, col, val, P);
    int k = orc_shifted_lop(d, x_set, r, sigma, nsig, seed, o);
    orc_dist_free(d);
    return k;
}

int orc_solve(int method, const orc_dist *d, double *x, double *r, orc_opts *o)
{
    switch (method) {
    case ORC_BICGSTAB:         return solve_plain(d, x, r, o);
    case ORC_CA_BICGSTAB:      return solve_ca(d, x, r, o);
    case ORC_PIPE_BICGSTAB:    return solve_pipe(d, x, r, o, 0);
    case ORC_PIPE_BICGSTAB_RR: return solve_pipe(d, x, r, o, 1);
    default: return -1;
    }
}

int orc_solve_coo(int method, int P, unsigned n, unsigned nnz, const unsigned *row,
                  const unsigned *col, const double *val, double *x, double *r, orc_opts *o)
{
    orc_dist *d = orc_dist_from_coo(n, nnz, row, col, val, P);
    int k = orc_solve(method, d, x, r, o);
    orc_dist_free(d);
    return k;
}

int orc_solve_coo_part(int method, int P, const int *counts, unsigned n, unsigned nnz, const unsigned *row,
                       const unsigned *col, const double *val, double *x, double *r, orc_opts *o)
{
    orc_dist *d = orc_dist_from_coo_part(n, nnz, row, col, val, P, counts);
    int k = orc_solve(method, d, x, r, o);
    orc_dist_free(d);
    return k;
}

void orc_spmv_coo_part(int P, const int *counts, unsigned n, unsigned nnz, const unsigned *row, const unsigned *col,
                       const double *val, const double *x, double *y)
{
    orc_dist *d = orc_dist_from_coo_part(n, nnz, row, col, val, P, counts);
    orc_spmv(d, x, y);
    orc_dist_free(d);
}

void orc_spmv_coo(int P, unsigned n, unsigned nnz, const unsigned *row, const unsigned *col,
                  const double *val, const double *x, double *y)
{
    orc_dist *d = orc_dist_from_coo(n, nnz, row, col, val, P);
    orc_spmv(d, x, y);
    orc_dist_free(d);
}
