/* bicg_mtx.c -- see bicg_mtx.h */
#define _GNU_SOURCE
#include "bicg_mtx.h"

#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include <ctype.h>
#include <stdint.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>

#ifdef BICG_HAVE_MPI
#include <mpi.h>
#endif

typedef struct { unsigned r, c; double v; } triplet;

/* BICG_MTX_VERBOSE=1: seconds per phase of the serial-mode loader on stderr */
static double wall(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1.0e-9 * (double)ts.tv_nsec;
}
static void phase(const char *what, double *t0)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("BICG_MTX_VERBOSE"); on = e && atoi(e); }
    const double t1 = wall();
    if (on) fprintf(stderr, "bicg_mtx: %-28s %.3f s\n", what, t1 - *t0);
    *t0 = t1;
}

typedef struct {
    unsigned long m, n, nz;
    int pattern, integer, symmetric;
    unsigned long long emitted;      /* entries handed to emit() by parse_entries (mirrored ones included) */
    unsigned long long lines;        /* entry lines parse_entries consumed */
    size_t data_off;      /* byte offset of the first entry line */
    char err[160];        /* parse_entries' message for a malformed line: printed by the caller once the error is final (a range
                           * that turns out to lie behind the nz-th entry line is never looked at, like in the reference) */
} mtx_header;

/* the reference's equal-rows partition, src/matrix.c:295-308 (kept local so that the loader does
 * not depend on the HIP library) */
static void partition(unsigned n, int nranks, int *counts, int *displs)
{
    const int base = (int)(n / (unsigned)nranks), extra = (int)(n % (unsigned)nranks);
    for (int p = 0; p < nranks; ++p) {
        counts[p] = base + (p < extra ? 1 : 0);
        displs[p] = p * base + (p < extra ? p : extra);
    }
}

static int owner_of(unsigned long row, unsigned long m, int nranks)
{
    const unsigned long base = m / (unsigned long)nranks, extra = m % (unsigned long)nranks;
    const unsigned long cut = extra * (base + 1);
    return (int)(row < cut ? row / (base + 1) : extra + (row - cut) / (base ? base : 1));
}

/* owner under an arbitrary contiguous partition: the rank p with displs[p] <= row < displs[p] + counts[p] */
static int owner_in(unsigned long row, const int *counts, const int *displs, int nranks)
{
    int lo = 0, hi = nranks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if ((unsigned long)displs[mid] <= row) lo = mid; else hi = mid - 1;
    }
    while (lo > 0 && counts[lo] == 0) --lo;      /* empty ranks share a displacement with their successor */
    while ((unsigned long)displs[lo] + (unsigned long)counts[lo] <= row && lo + 1 < nranks) ++lo;
    return lo;
}

/* Contiguous row blocks with (nearly) equal numbers of NON-ZEROS: cut k is placed where the running
 * count is closest to k/P of the total. The idea is the reference's abandoned DYNAMIC_ROWS branch
 * (archive/matrix.c:407-446: each rank takes rows until it holds nz/P entries, the last rank takes
 * the rest); cutting against the cumulative targets instead keeps the error from piling up on the
 * last rank, and every rank gets at least one row when n >= P. */
void bicg_partition_nnz(const unsigned int *row_nnz, unsigned int n, int nranks, int *counts, int *displs)
{
    unsigned long long total = 0, run = 0;
    for (unsigned i = 0; i < n; ++i) total += row_nnz[i];
    unsigned row = 0;
    displs[0] = 0;
    for (int k = 1; k < nranks; ++k) {
        const unsigned long long target = total * (unsigned long long)k / (unsigned long long)nranks;
        const unsigned min_row = (unsigned)displs[k - 1] + (n >= (unsigned)nranks ? 1u : 0u);
        const unsigned max_row = n >= (unsigned)nranks ? n - (unsigned)(nranks - k) : n;
        while (row < n && run + row_nnz[row] <= target) run += row_nnz[row++];
        /* row = first row whose inclusion would exceed the target: take it if that is closer */
        if (row < n && run < target && (run + row_nnz[row]) - target < target - run) run += row_nnz[row++];
        while (row < min_row && row < n) run += row_nnz[row++];
        while (row > max_row) run -= row_nnz[--row];
        displs[k] = (int)row;
    }
    for (int k = 0; k < nranks; ++k) counts[k] = (k + 1 < nranks ? displs[k + 1] : (int)n) - displs[k];
}

/* worker threads of the loader: BICG_MTX_THREADS, else the cores this process may use divided by the ranks that share them
 * (at most 32 per rank: the cap is applied AFTER the division -- 8 ranks on a 256-thread host get 32 each, not 2);
 * small inputs stay on the calling thread */
static long loader_threads_shared(size_t work_bytes, int ranks_on_node)
{
    long nt = sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) nt = CPU_COUNT(&set);
    nt /= ranks_on_node > 0 ? ranks_on_node : 1;
    if (nt > 32) nt = 32;
    if (work_bytes < ((size_t)1 << 22)) nt = 1;
    const char *env = getenv("BICG_MTX_THREADS");
    if (env) nt = atol(env);
    return nt < 1 ? 1 : nt;
}
static long loader_threads(size_t work_bytes) { return loader_threads_shared(work_bytes, 1); }
/* fn(arg + t * stride) on nt threads (the last one on the caller's); a thread that cannot be created runs inline */
static void run_threads(long nt, void *(*fn)(void *), void *args, size_t stride)
{
    pthread_t *tid = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
    for (long t = 0; t + 1 < nt; ++t)
        if (pthread_create(&tid[t], NULL, fn, (char *)args + (size_t)t * stride) != 0) { fn((char *)args + (size_t)t * stride); tid[t] = 0; }
    fn((char *)args + (size_t)(nt - 1) * stride);
    for (long t = 0; t + 1 < nt; ++t) if (tid[t]) pthread_join(tid[t], NULL);
    free(tid);
}

typedef struct { int fd; char *dst; size_t off, len, got; } read_job;
static void *read_job_run(void *arg)
{
    read_job *j = (read_job *)arg;
    j->got = 0;
    while (j->got < j->len) {
        const ssize_t r = pread(j->fd, j->dst + j->got, j->len - j->got, (off_t)(j->off + j->got));
        if (r <= 0) break;
        j->got += (size_t)r;
    }
    return NULL;
}

static char *slurp_range(const char *path, size_t off, size_t len, size_t *got)
{
    if (len == (size_t)-1 && off == 0) {        /* the whole file: byte ranges read (and their pages touched) by several threads */
        struct stat st;
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return NULL;
        if (fstat(fd, &st) != 0) { close(fd); return NULL; }
        const size_t sz = (size_t)st.st_size;
        char *buf = (char *)malloc(sz + 1);
        if (!buf) { close(fd); return NULL; }
        long nt = loader_threads(sz);
        read_job *jobs = (read_job *)calloc((size_t)nt, sizeof(read_job));
        for (long t = 0; t < nt; ++t) {
            const size_t a = sz * (size_t)t / (size_t)nt, b = sz * (size_t)(t + 1) / (size_t)nt;
            jobs[t].fd = fd; jobs[t].dst = buf + a; jobs[t].off = a; jobs[t].len = b - a;
        }
        run_threads(nt, read_job_run, jobs, sizeof(read_job));
        size_t rd = 0;
        for (long t = 0; t < nt; ++t) { rd += jobs[t].got; if (jobs[t].got < jobs[t].len) break; }   /* a short range ends the data */
        free(jobs);
        close(fd);
        buf[rd] = 0;
        *got = rd;
        return buf;
    }
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    if (len == (size_t)-1) {
        fseek(f, 0, SEEK_END);
        len = (size_t)ftell(f) - off;
    }
    fseek(f, (long)off, SEEK_SET);
    char *buf = (char *)malloc(len + 1);
    if (!buf) { fclose(f); return NULL; }
    const size_t rd = fread(buf, 1, len, f);
    fclose(f);
    buf[rd] = 0;
    *got = rd;
    return buf;
}

static size_t file_size(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END);
    const size_t sz = (size_t)ftell(f);
    fclose(f);
    return sz;
}

/* banner + size line; returns 0 on success */
static int parse_header(const char *buf, mtx_header *h)
{
    const char *p = buf;
    h->err[0] = 0; h->lines = 0;
    if (strncmp(p, "%%MatrixMarket", 14) != 0) { fprintf(stderr, "ERROR: Could not process Matrix Market banner.\n"); return 2; }
    const char *eol = strchr(p, '\n');
    if (!eol) return 2;
    char banner[1100];
    size_t bl = (size_t)(eol - p);
    if (bl > sizeof banner - 1) bl = sizeof banner - 1;
    for (size_t i = 0; i < bl; ++i) banner[i] = (char)tolower((unsigned char)p[i]);
    banner[bl] = 0;
    h->pattern = strstr(banner, "pattern") != NULL;
    h->integer = strstr(banner, "integer") != NULL;
    h->symmetric = strstr(banner, "symmetric") != NULL && strstr(banner, "skew") == NULL;
    h->emitted = 0;
    /* skew-symmetric / hermitian storage would need sign / conjugate handling on the mirrored half: refuse
     * rather than silently treat the file as general (the reference's block loader ignores the banner
     * altogether, src/matrix.c:268-396) */
    if (!strstr(banner, "coordinate") || strstr(banner, "complex") || strstr(banner, "skew") || strstr(banner, "hermitian")) {
        fprintf(stderr, "Sorry, this application does not support Market Market type: [%s]\n", banner);
        return 3;
    }
    p = eol + 1;
    while (*p == '%') { p = strchr(p, '\n'); if (!p) return 4; ++p; }
    char *q;
    h->m = strtoul(p, &q, 10); p = q;
    h->n = strtoul(p, &q, 10); p = q;
    h->nz = strtoul(p, &q, 10); p = q;
    if (!h->m || !h->n) { fprintf(stderr, "ERROR: Could not read matrix size.\n"); return 5; }
    const char *nl = strchr(p, '\n');
    h->data_off = nl ? (size_t)(nl + 1 - buf) : (size_t)(p - buf);
    return 0;
}

typedef struct { triplet *t; size_t n, cap; } tvec;
static void tpush(tvec *v, unsigned r, unsigned c, double val)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->t = (triplet *)realloc(v->t, sizeof(triplet) * v->cap); }
    v->t[v->n].r = r; v->t[v->n].c = c; v->t[v->n].v = val; v->n++;
}

/* ---- number conversion. The reference reads an entry with fscanf("%d %d %lg") (src/matrix.c:333, 366): correctly
 * rounded doubles. strtoul / strtod give the same values but cost 130 ns per line (two thirds of it strtod on 17-digit
 * values); the two routines below handle the ordinary spellings and hand everything else back to libc:
 *   fast_uint    plain digit strings;
 *   fast_double  [sign] digits [. digits] [e [sign] digits] with at most 19 significant digits, converted with the
 *                Eisel-Lemire algorithm (D. Lemire, "Number parsing at a gigabyte per second", 2021: one 64 x 128-bit
 *                product with a truncated 128-bit power of ten, bicg_pow10_table.h, decides all but a sliver of inputs
 *                exactly; the sliver -- half-way ambiguity, subnormals, overflow -- returns "not handled"). The result
 *                is the correctly rounded double or no result at all; tests/test_host_loader.py compares it with
 *                Python's float() on several hundred thousand strings, half-way neighbourhoods included. */
#include "bicg_pow10_table.h"

static inline int fast_uint(const char *p, unsigned long *out, const char **endp)
{
    if ((unsigned)(*p - '0') > 9u) return 0;
    unsigned long v = 0;
    int nd = 0;
    while ((unsigned)(*p - '0') <= 9u) { v = v * 10u + (unsigned long)(*p - '0'); ++p; if (++nd > 18) return 0; }
    *out = v; *endp = p;
    return 1;
}

static int eisel_lemire(uint64_t man, int exp10, int neg, double *out)
{
    if (exp10 < BICG_POW10_MIN || exp10 > BICG_POW10_MAX) return 0;
    const int clz = __builtin_clzll(man);
    man <<= clz;
    uint64_t exp2 = (uint64_t)(((217706 * exp10) >> 16) + 64 + 1023) - (uint64_t)clz;     /* floor(exp10 * log2(10)) + bias */
    const unsigned long long *pw = bicg_pow10_128[exp10 - BICG_POW10_MIN];
    unsigned __int128 x = (unsigned __int128)man * pw[1];
    uint64_t xhi = (uint64_t)(x >> 64), xlo = (uint64_t)x;
    if ((xhi & 0x1FF) == 0x1FF && xlo + man < man) {          /* the truncated low half of the power could matter: use it */
        unsigned __int128 y = (unsigned __int128)man * pw[0];
        const uint64_t yhi = (uint64_t)(y >> 64), ylo = (uint64_t)y;
        uint64_t mhi = xhi, mlo = xlo + yhi;
        if (mlo < xlo) ++mhi;
        if ((mhi & 0x1FF) == 0x1FF && mlo + 1 == 0 && ylo + man < man) return 0;
        xhi = mhi; xlo = mlo;
    }
    const uint64_t msb = xhi >> 63;
    uint64_t m = xhi >> (msb + 9);                            /* 54 bits */
    exp2 -= 1 ^ msb;
    if (xlo == 0 && (xhi & 0x1FF) == 0 && (m & 3) == 1) return 0;     /* exactly half way between two doubles? not decided here */
    m += m & 1;
    m >>= 1;
    if (m >> 53) { m >>= 1; ++exp2; }
    if (exp2 - 1 >= 0x7FF - 1) return 0;                       /* subnormal, zero, infinity: libc */
    uint64_t bits = exp2 << 52 | (m & 0x000FFFFFFFFFFFFFull);
    if (neg) bits |= 0x8000000000000000ull;
    memcpy(out, &bits, sizeof bits);
    return 1;
}

static int fast_double(const char *p, double *out, const char **endp)
{
    while (*p == ' ' || *p == '\t') ++p;
    int neg = 0;
    if (*p == '-') { neg = 1; ++p; } else if (*p == '+') ++p;
    if (p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) return 0;        /* hexadecimal floats: libc */
    uint64_t w = 0;
    int nsig = 0, ndig = 0, exp10 = 0, dropped = 0;       /* digits beyond the 19th are dropped: the value lies in [w, w + 1) x 10^exp10 */
    for (; (unsigned)(*p - '0') <= 9u; ++p, ++ndig) {
        if (nsig == 0 && *p == '0') continue;                           /* leading zeros */
        if (nsig == 19) { ++exp10; dropped |= *p != '0'; if (exp10 > 100000) return 0; continue; }
        ++nsig;
        w = w * 10u + (uint64_t)(*p - '0');
    }
    if (*p == '.') {
        ++p;
        for (; (unsigned)(*p - '0') <= 9u; ++p, ++ndig) {
            if (nsig == 19) { dropped |= *p != '0'; continue; }
            if (exp10 < -100000) return 0;
            --exp10;
            if (nsig == 0 && *p == '0') continue;
            ++nsig;
            w = w * 10u + (uint64_t)(*p - '0');
        }
    }
    if (ndig == 0) return 0;                                           /* "inf", "nan", ".", garbage: libc decides */
    if (*p == 'e' || *p == 'E') {
        const char *e = p + 1;
        int eneg = 0, ev = 0, nd = 0;
        if (*e == '-') { eneg = 1; ++e; } else if (*e == '+') ++e;
        for (; (unsigned)(*e - '0') <= 9u; ++e, ++nd) if (ev < 100000) ev = ev * 10 + (*e - '0');
        if (nd) { exp10 += eneg ? -ev : ev; p = e; }                    /* "1e" / "1e+": the number ends before the e, like strtod */
    }
    if (w == 0) { *out = neg ? -0.0 : 0.0; *endp = p; return 1; }
    if (dropped) {        /* more than 19 digits: decided when both ends of [w, w + 1) round to the same double */
        double lo, hi;
        if (!eisel_lemire(w, exp10, neg, &lo) || !eisel_lemire(w + 1, exp10, neg, &hi) || memcmp(&lo, &hi, sizeof lo) != 0) return 0;
        *out = lo; *endp = p;
        return 1;
    }
    if (w < (1ull << 53) && exp10 >= -22 && exp10 <= 22) {
        /* both factors are exact doubles: ONE correctly rounded operation (Clinger 1990). Eisel-Lemire below declines
         * exactly these -- a truncated power of ten puts e.g. 20 x 10^-1 a hair under 2 */
        static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                       1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        double d = (double)w;
        d = exp10 < 0 ? d / p10[-exp10] : d * p10[exp10];
        *out = neg ? -d : d;
        *endp = p;
        return 1;
    }
    if (!eisel_lemire(w, exp10, neg, out)) return 0;
    *endp = p;                                  /* (only on success: the caller hands the SAME text to strtod otherwise) */
    return 1;
}

/* the conversions as the tokeniser uses them, for the tests: 0 = converted by the routines above, 1 = handed to strtod */
int bicg_mtx_parse_double(const char *text, double *value, int *consumed)
{
    const char *end = text;
    if (fast_double(text, value, &end)) { *consumed = (int)(end - text); return 0; }
    char *q;
    *value = strtod(text, &q);
    *consumed = (int)(q - text);
    return 1;
}

/* tokenise entry lines in [p, end): calls emit(row, col, val) with 0-based GLOBAL indices (src/matrix.c:333-334) */
static int parse_entries(const char *p, const char *end, mtx_header *h, unsigned long max_entries,
                         void (*emit)(void *, unsigned long, unsigned long, double), void *ctx)
{
    char *q;
    unsigned long seen = 0;
    h->emitted = 0;
    while (p < end && seen < max_entries) {
        while (p < end && isspace((unsigned char)*p)) ++p;
        if (p >= end) break;
        if (*p == '%') { while (p < end && *p != '\n') ++p; continue; }
        unsigned long i, j;
        if (!fast_uint(p, &i, &p)) {
            i = strtoul(p, &q, 10);
            if (q == p) { snprintf(h->err, sizeof h->err, "ERROR: reading matrix data.\n"); return 6; }
            p = q;
        }
        while (*p == ' ' || *p == '\t') ++p;
        if (!fast_uint(p, &j, &p)) { j = strtoul(p, &q, 10); p = q; }
        double v = 1.0;
        if (!h->pattern && !fast_double(p, &v, &p)) { v = strtod(p, &q); p = q; }
        --i; --j;
        if (i >= h->m || j >= h->n) {    /* (0 wraps around: also caught) the reference would index outside its arrays */
            snprintf(h->err, sizeof h->err, "ERROR: matrix entry (%lu, %lu) lies outside the %lu x %lu matrix.\n", i + 1, j + 1, (unsigned long)h->m, (unsigned long)h->n);
            return 7;
        }
        emit(ctx, i, j, v);
        h->emitted++;
        if (h->symmetric && i != j) { emit(ctx, j, i, v); h->emitted++; }
        ++seen;
    }
    h->lines = seen;
    return 0;
}

static void fill_info(const mtx_header *h, int nranks, INFO_Matrix *info)
{   /* equal-rows partition; the nnz-balanced loaders overwrite recvcounts/displs afterwards */
    info->rows = (unsigned)h->m; info->cols = (unsigned)h->n; info->nz = (unsigned)h->nz;
    memcpy(info->code, h->pattern ? "MCPG" : (h->integer ? "MCIG" : "MCRG"), 4);
    if (h->symmetric) info->code[3] = 'S';
    info->recvcounts = (int *)malloc(sizeof(int) * (size_t)nranks);
    info->displs = (int *)malloc(sizeof(int) * (size_t)nranks);
    partition((unsigned)h->m, nranks, info->recvcounts, info->displs);
}

static bicg_block_builder_fn g_builder = NULL;
void bicg_mtx_set_block_builder(bicg_block_builder_fn fn) { g_builder = fn; }

/* The triplets of rows [lo, hi) arrive as a few lists ("segments") whose concatenation is file order. Assembly into the
 * two CSR blocks is split by ROW range: thread k owns local rows [r0, r1), walks every segment twice (count, then
 * scatter) and keeps what falls into its rows -- no shared counters, no histogram per thread, and the arrival (= file)
 * order inside a row survives because every thread walks the segments in order. 24 M triplets are 383 MB: the passes
 * stream at memory speed. */
typedef struct { triplet *t; size_t n; } tseg;
typedef struct { const tseg *seg; int nseg; unsigned lo, hi, r0, r1; CSR_Matrix *d, *o; unsigned *cur_d, *cur_o; int pass; } asm_job;
static void *asm_job_run(void *arg)
{
    asm_job *j = (asm_job *)arg;
    const unsigned lo = j->lo, hi = j->hi, a = j->lo + j->r0, b = j->lo + j->r1;
    if (j->pass == 1) {
        for (unsigned r = j->r0; r < j->r1; ++r) { j->cur_d[r] = j->d->ptr[r]; j->cur_o[r] = j->o->ptr[r]; }
    }
    for (int s = 0; s < j->nseg; ++s) {
        const triplet *t = j->seg[s].t;
        const size_t n = j->seg[s].n;
        if (j->pass == 0) {
            for (size_t e = 0; e < n; ++e) {
                if (t[e].r < a || t[e].r >= b) continue;
                if (t[e].c >= lo && t[e].c < hi) j->d->ptr[t[e].r - lo + 1]++; else j->o->ptr[t[e].r - lo + 1]++;
            }
        } else {
            for (size_t e = 0; e < n; ++e) {
                if (t[e].r < a || t[e].r >= b) continue;
                const unsigned r = t[e].r - lo;
                if (t[e].c >= lo && t[e].c < hi) { const unsigned k = j->cur_d[r]++; j->d->col[k] = t[e].c - lo; j->d->val[k] = t[e].v; }
                else { const unsigned k = j->cur_o[r]++; j->o->col[k] = t[e].c; j->o->val[k] = t[e].v; }
            }
        }
    }
    return NULL;
}

static void build_blocks(const tseg *seg, int nseg, unsigned lo, unsigned hi, unsigned ncols, CSR_Matrix *diag, CSR_Matrix *offd)
{
    size_t nt_all = 0;
    for (int s = 0; s < nseg; ++s) nt_all += seg[s].n;
    if (g_builder) {            /* e.g. bicg_coo_to_blocks_device: sort / scan / scatter on the GPU */
        unsigned *r = (unsigned *)malloc(sizeof(unsigned) * (nt_all ? nt_all : 1)), *c = (unsigned *)malloc(sizeof(unsigned) * (nt_all ? nt_all : 1));
        double *v = (double *)malloc(sizeof(double) * (nt_all ? nt_all : 1));
        size_t at = 0;
        for (int s = 0; s < nseg; ++s)
            for (size_t e = 0; e < seg[s].n; ++e, ++at) { r[at] = seg[s].t[e].r; c[at] = seg[s].t[e].c; v[at] = seg[s].t[e].v; }
        const int rc = g_builder(r, c, v, (unsigned long)nt_all, lo, hi, ncols, diag, offd);
        free(r); free(c); free(v);
        if (rc == 0) return;
        fprintf(stderr, "bicg_mtx: block builder failed (%d), using the host path\n", rc);
    }
    const unsigned rows = hi - lo;
    diag->rows = rows; diag->cols = rows;        /* cols = local rows, src/matrix.c:343-345 */
    offd->rows = rows; offd->cols = ncols;       /* cols = n,          src/matrix.c:350-352 */
    diag->ptr = (unsigned *)calloc((size_t)rows + 1, sizeof(unsigned));
    offd->ptr = (unsigned *)calloc((size_t)rows + 1, sizeof(unsigned));
    unsigned *cur = (unsigned *)malloc(sizeof(unsigned) * 2 * ((size_t)rows + 1));
    long nt = loader_threads(nt_all * sizeof(triplet));
    if ((unsigned)nt > rows) nt = rows ? (long)rows : 1;
    asm_job *jobs = (asm_job *)calloc((size_t)nt, sizeof(asm_job));
    for (long t = 0; t < nt; ++t) {
        jobs[t].seg = seg; jobs[t].nseg = nseg; jobs[t].lo = lo; jobs[t].hi = hi;
        jobs[t].r0 = (unsigned)((unsigned long long)rows * (unsigned long long)t / (unsigned long long)nt);
        jobs[t].r1 = (unsigned)((unsigned long long)rows * (unsigned long long)(t + 1) / (unsigned long long)nt);
        jobs[t].d = diag; jobs[t].o = offd; jobs[t].cur_d = cur; jobs[t].cur_o = cur + rows + 1; jobs[t].pass = 0;
    }
    run_threads(nt, asm_job_run, jobs, sizeof(asm_job));                      /* entries per row */
    for (unsigned i = 0; i < rows; ++i) { diag->ptr[i + 1] += diag->ptr[i]; offd->ptr[i + 1] += offd->ptr[i]; }
    const size_t nd = diag->ptr[rows], no = offd->ptr[rows];
    diag->nz = (unsigned)nd; offd->nz = (unsigned)no;
    diag->col = (unsigned *)malloc(sizeof(unsigned) * (nd ? nd : 1)); diag->val = (double *)malloc(sizeof(double) * (nd ? nd : 1));
    offd->col = (unsigned *)malloc(sizeof(unsigned) * (no ? no : 1)); offd->val = (double *)malloc(sizeof(double) * (no ? no : 1));
    for (long t = 0; t < nt; ++t) jobs[t].pass = 1;
    run_threads(nt, asm_job_run, jobs, sizeof(asm_job));                      /* arrival (= file) order inside every row */
    free(jobs); free(cur);
}

typedef struct { tvec mine; unsigned lo, hi; } serial_ctx;
static void emit_serial(void *c, unsigned long i, unsigned long j, double v)
{
    serial_ctx *s = (serial_ctx *)c;
    if (i >= s->lo && i < s->hi) tpush(&s->mine, (unsigned)i, (unsigned)j, v);
}

/* The entry lines tokenised by several threads: the text is cut into byte ranges at line boundaries, every thread keeps
 * the triplets of [lo, hi) it finds in its own list, and the lists -- in range order: file order, like the one-thread
 * pass -- go to build_blocks as they are (the reference has every rank fscanf() the whole file twice,
 * src/matrix.c:315-341, 357-393; a Transport-sized file -- 840 MB, 23.9 M lines -- took the single tokeniser 2.1 s of a
 * 2.9 s run). BICG_MTX_THREADS overrides the count (default: the cores the process may use, at most 32; 1 = serial). */
typedef struct { const char *p, *end; mtx_header h; serial_ctx s; int rc; unsigned long max_entries; } parse_job;
static void *parse_job_run(void *arg)
{
    parse_job *j = (parse_job *)arg;
    j->rc = parse_entries(j->p, j->end, &j->h, j->max_entries, emit_serial, &j->s);
    return NULL;
}
/* on success *segs (malloc'ed, *nseg lists, each malloc'ed) holds the triplets of rows [lo, hi) in file order */
static int parse_threaded(const char *p, const char *end, mtx_header *h, unsigned lo, unsigned hi, long threads, tseg **segs, int *nseg)
{
    long nt = threads > 0 ? threads : loader_threads((size_t)(end - p));
    parse_job *jobs = (parse_job *)calloc((size_t)nt, sizeof(parse_job));
    const size_t len = (size_t)(end - p);
    const char *cut = p;
    for (long t = 0; t < nt; ++t) {
        const char *stop = t == nt - 1 ? end : p + len * (size_t)(t + 1) / (size_t)nt;
        if (stop < cut) stop = cut;                                        /* the previous range ran past this one's end: empty */
        while (stop < end && stop > cut && stop[-1] != '\n') ++stop;      /* a range ends after a newline */
        jobs[t].p = cut; jobs[t].end = stop; jobs[t].h = *h; jobs[t].s.lo = lo; jobs[t].s.hi = hi;
        jobs[t].max_entries = nt == 1 ? h->nz : (unsigned long)-1;        /* one thread: stop after the banner's count like the reference */
        cut = stop;
    }
    run_threads(nt, parse_job_run, jobs, sizeof(parse_job));
    int rc = 0;
    /* The reference reads exactly the banner's nz entry lines (src/matrix.c:315: fewer is an error, more are never looked at).
     * Several ranges: lines past the nz-th are dropped -- the range that holds the nz-th line is parsed again up to it -- so the
     * loaded matrix does not depend on the number of threads. */
    {
        unsigned long long before = 0;
        for (long t = 0; t < nt && !rc; ++t) {
            if (before >= h->nz) {                                /* entirely past the last entry: never looked at, whatever it holds */
                free(jobs[t].s.mine.t); memset(&jobs[t].s.mine, 0, sizeof jobs[t].s.mine);
                jobs[t].h.emitted = 0; jobs[t].h.lines = 0; jobs[t].rc = 0;
                continue;
            }
            if (h->nz != (unsigned long)-1 && (jobs[t].rc || before + jobs[t].h.lines > h->nz)) {
                /* the range that holds the nz-th line (or a malformed line that may lie behind it): read again, up to the
                 * nz-th line only, and judge THAT pass */
                free(jobs[t].s.mine.t); memset(&jobs[t].s.mine, 0, sizeof jobs[t].s.mine);
                jobs[t].h = *h; jobs[t].h.emitted = 0; jobs[t].h.lines = 0;
                jobs[t].max_entries = (unsigned long)(h->nz - before);
                parse_job_run(&jobs[t]);
            }
            if (jobs[t].rc) {     /* (a byte range of the MPI mode, nz unknown: the caller decides whether the line counts -- it may lie behind the nz-th) */
                rc = jobs[t].rc;
                if (h->nz != (unsigned long)-1) fputs(jobs[t].h.err, stderr);
                memcpy(h->err, jobs[t].h.err, sizeof h->err);
                break;
            }
            before += jobs[t].h.lines;
        }
        if (!rc && h->nz != (unsigned long)-1 && before < h->nz) { fprintf(stderr, "ERROR: reading matrix data.\n"); rc = 6; }     /* the reference's message, src/matrix.c:318 */
        h->lines = before < h->nz ? before : h->nz;
    }
    h->emitted = 0;
    *segs = (tseg *)calloc((size_t)nt, sizeof(tseg));
    *nseg = (int)nt;
    for (long t = 0; t < nt; ++t) {
        h->emitted += jobs[t].h.emitted;
        (*segs)[t].t = jobs[t].s.mine.t; (*segs)[t].n = jobs[t].s.mine.n;
    }
    free(jobs);
    if (rc) { for (int i = 0; i < *nseg; ++i) free((*segs)[i].t); free(*segs); *segs = NULL; *nseg = 0; }
    return rc;
}

int bicg_mtx_load_block_part(const char *path, int rank, int nranks, int part, CSR_Matrix *diag, CSR_Matrix *offd,
                             INFO_Matrix *info)
{
    size_t len = 0;
    double t0 = wall();
    char *buf = slurp_range(path, 0, (size_t)-1, &len);
    if (!buf) { fprintf(stderr, "ERROR: can't open file \"%s\"\n", path); return 1; }
    phase("read the file", &t0);
    mtx_header h;
    int rc = parse_header(buf, &h);
    if (rc) { free(buf); return rc; }
    fill_info(&h, nranks, info);
    tseg *segs = NULL;
    int nseg = 0;
    t0 = wall();
    if (part == BICG_PART_NNZ) {
        /* the cuts depend on every row's length: the text is still tokenised ONCE -- all rows are kept, counted from the
         * lists, and build_blocks picks this rank's range out of them */
        rc = parse_threaded(buf + h.data_off, buf + len, &h, 0u, (unsigned)h.m, 0, &segs, &nseg);
        if (!rc) {
            unsigned *cnt = (unsigned *)calloc(h.m ? h.m : 1, sizeof(unsigned));
            for (int sg = 0; sg < nseg; ++sg)
                for (size_t e = 0; e < segs[sg].n; ++e) cnt[segs[sg].t[e].r]++;
            bicg_partition_nnz(cnt, (unsigned)h.m, nranks, info->recvcounts, info->displs);
            free(cnt);
        }
    }
    const unsigned lo = (unsigned)info->displs[rank], hi = (unsigned)(info->displs[rank] + info->recvcounts[rank]);
    if (part != BICG_PART_NNZ) rc = parse_threaded(buf + h.data_off, buf + len, &h, lo, hi, 0, &segs, &nseg);
    phase("tokenise", &t0);
    free(buf);
    if (rc) return rc;
    /* nz = entries of the matrix that is actually solved: for 'symmetric' storage the banner counts one
     * triangle only (the reference's loader does not mirror: it would solve that triangle) */
    info->nz = (unsigned)h.emitted;
    if (h.symmetric && rank == 0)
        fprintf(stderr, "bicg_mtx: 'symmetric' storage: mirrored to %llu entries (the reference's block loader keeps the stored triangle only)\n",
                h.emitted);
    t0 = wall();
    build_blocks(segs, nseg, lo, hi, (unsigned)h.n, diag, offd);
    phase("diag / offd CSR blocks", &t0);
    for (int i = 0; i < nseg; ++i) free(segs[i].t);
    free(segs);
    return 0;
}

int bicg_mtx_load_block(const char *path, int rank, int nranks, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
{
    return bicg_mtx_load_block_part(path, rank, nranks, BICG_PART_ROWS, diag, offd, info);
}

/* ------------------------------------------------------------------------------------------
 * Binary block cache: the parsed blocks of one rank, so that the next run skips the text file.
 *   header | recvcounts[P] | displs[P] | diag ptr,col,val | offd ptr,col,val | FNV-1a of all of it
 * A cache entry is only used when it was written for the same (rank, nranks, partition mode) and the
 * source file still has the size and modification time recorded in the header.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    char     magic[8];              /* "BICGBLK1" */
    uint32_t version, nranks, rank, part;
    uint32_t rows, cols, nz;        /* global, as in the banner */
    char     code[4];
    uint32_t local_rows, nnz_d, nnz_o, offd_cols;
    uint64_t src_size;
    int64_t  src_mtime;
} cache_header;

static uint64_t fnv1a(uint64_t h, const void *data, size_t n)
{
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

static int src_stamp(const char *src, uint64_t *size, int64_t *mtime)
{
    struct stat st;
    if (!src || stat(src, &st) != 0) { *size = 0; *mtime = 0; return 1; }
    *size = (uint64_t)st.st_size; *mtime = (int64_t)st.st_mtime;
    return 0;
}

int bicg_mtx_cache_save(const char *cache_path, const char *src_path, int rank, int nranks, int part,
                        const CSR_Matrix *diag, const CSR_Matrix *offd, const INFO_Matrix *info)
{
    cache_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "BICGBLK1", 8);
    h.version = 1; h.nranks = (uint32_t)nranks; h.rank = (uint32_t)rank; h.part = (uint32_t)part;
    h.rows = info->rows; h.cols = info->cols; h.nz = info->nz;
    memcpy(h.code, info->code, 4);
    h.local_rows = diag->rows; h.nnz_d = diag->ptr[diag->rows]; h.nnz_o = offd->ptr ? offd->ptr[offd->rows] : 0u;
    h.offd_cols = offd->cols;
    (void)src_stamp(src_path, &h.src_size, &h.src_mtime);
    char tmp[4096];
    snprintf(tmp, sizeof tmp, "%s.tmp%d", cache_path, rank);
    FILE *f = fopen(tmp, "wb");
    if (!f) return 1;
    uint64_t sum = 1469598103934665603ull;
    int bad = 0;
#define PUT(ptr, bytes) do { const size_t nb_ = (bytes); if (nb_ && fwrite((ptr), 1, nb_, f) != nb_) bad = 1; sum = fnv1a(sum, (ptr), nb_); } while (0)
    PUT(&h, sizeof h);
    PUT(info->recvcounts, sizeof(int) * (size_t)nranks);
    PUT(info->displs, sizeof(int) * (size_t)nranks);
    PUT(diag->ptr, sizeof(unsigned) * ((size_t)h.local_rows + 1));
    PUT(diag->col, sizeof(unsigned) * (size_t)h.nnz_d);
    PUT(diag->val, sizeof(double) * (size_t)h.nnz_d);
    PUT(offd->ptr, sizeof(unsigned) * ((size_t)h.local_rows + 1));
    PUT(offd->col, sizeof(unsigned) * (size_t)h.nnz_o);
    PUT(offd->val, sizeof(double) * (size_t)h.nnz_o);
#undef PUT
    if (fwrite(&sum, 1, sizeof sum, f) != sizeof sum) bad = 1;
    if (fclose(f) != 0) bad = 1;
    if (bad || rename(tmp, cache_path) != 0) { remove(tmp); return 1; }   /* readers never see a half-written file */
    return 0;
}

int bicg_mtx_cache_load(const char *cache_path, const char *src_path, int rank, int nranks, int part,
                        CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
{
    FILE *f = fopen(cache_path, "rb");
    if (!f) return 1;
    cache_header h;
    uint64_t sum = 1469598103934665603ull, want = 0, ssize = 0;
    int64_t smtime = 0;
    int rc = 2;
    memset(diag, 0, sizeof *diag); memset(offd, 0, sizeof *offd); memset(info, 0, sizeof *info);
    if (fread(&h, 1, sizeof h, f) != sizeof h) goto out;
    if (memcmp(h.magic, "BICGBLK1", 8) != 0 || h.version != 1) goto out;
    if ((int)h.nranks != nranks || (int)h.rank != rank || (int)h.part != part) goto out;
    if (src_path && (src_stamp(src_path, &ssize, &smtime) != 0 || ssize != h.src_size || smtime != h.src_mtime)) goto out;
    sum = fnv1a(sum, &h, sizeof h);
    info->rows = h.rows; info->cols = h.cols; info->nz = h.nz;
    memcpy(info->code, h.code, 4);
    info->recvcounts = (int *)malloc(sizeof(int) * (size_t)nranks);
    info->displs = (int *)malloc(sizeof(int) * (size_t)nranks);
    diag->rows = h.local_rows; diag->cols = h.local_rows; diag->nz = h.nnz_d;
    offd->rows = h.local_rows; offd->cols = h.offd_cols; offd->nz = h.nnz_o;
    diag->ptr = (unsigned *)malloc(sizeof(unsigned) * ((size_t)h.local_rows + 1));
    diag->col = (unsigned *)malloc(sizeof(unsigned) * (h.nnz_d ? h.nnz_d : 1));
    diag->val = (double *)malloc(sizeof(double) * (h.nnz_d ? h.nnz_d : 1));
    offd->ptr = (unsigned *)malloc(sizeof(unsigned) * ((size_t)h.local_rows + 1));
    offd->col = (unsigned *)malloc(sizeof(unsigned) * (h.nnz_o ? h.nnz_o : 1));
    offd->val = (double *)malloc(sizeof(double) * (h.nnz_o ? h.nnz_o : 1));
    rc = 3;
#define GET(ptr, bytes) do { const size_t nb_ = (bytes); if (nb_ && fread((ptr), 1, nb_, f) != nb_) goto out; sum = fnv1a(sum, (ptr), nb_); } while (0)
    GET(info->recvcounts, sizeof(int) * (size_t)nranks);
    GET(info->displs, sizeof(int) * (size_t)nranks);
    GET(diag->ptr, sizeof(unsigned) * ((size_t)h.local_rows + 1));
    GET(diag->col, sizeof(unsigned) * (size_t)h.nnz_d);
    GET(diag->val, sizeof(double) * (size_t)h.nnz_d);
    GET(offd->ptr, sizeof(unsigned) * ((size_t)h.local_rows + 1));
    GET(offd->col, sizeof(unsigned) * (size_t)h.nnz_o);
    GET(offd->val, sizeof(double) * (size_t)h.nnz_o);
#undef GET
    if (fread(&want, 1, sizeof want, f) != sizeof want || want != sum) goto out;
    if (diag->ptr[h.local_rows] != h.nnz_d || offd->ptr[h.local_rows] != h.nnz_o) goto out;
    rc = 0;
out:
    fclose(f);
    if (rc != 0 && rc != 1) {          /* partial read: release whatever was allocated */
        free(diag->ptr); free(diag->col); free(diag->val);
        free(offd->ptr); free(offd->col); free(offd->val);
        free(info->recvcounts); free(info->displs);
        memset(diag, 0, sizeof *diag); memset(offd, 0, sizeof *offd); memset(info, 0, sizeof *info);
    }
    return rc;
}

#ifdef BICG_HAVE_MPI
int bicg_mtx_load_block_mpi(const char *path, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
{
    return bicg_mtx_load_block_mpi_part(path, BICG_PART_ROWS, diag, offd, info);
}

int bicg_mtx_load_block_mpi_part(const char *path, int part, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
{
    int np = 1, me = 0;
    MPI_Comm_size(MPI_COMM_WORLD, &np);
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
    /* header: every rank reads the first 64 KiB */
    size_t got = 0;
    char *head = slurp_range(path, 0, 65536, &got);
    if (!head) { fprintf(stderr, "ERROR: can't open file \"%s\"\n", path); return 1; }
    mtx_header h;
    int rc = parse_header(head, &h);
    free(head);
    if (rc) return rc;
    fill_info(&h, np, info);

    /* this rank's byte range of the entry lines, widened to whole lines: a range owns the lines
     * that START inside it */
    const size_t fsz = file_size(path), body = fsz - h.data_off;
    size_t a = h.data_off + (size_t)((double)body * me / np), b = h.data_off + (size_t)((double)body * (me + 1) / np);
    if (me == np - 1) b = fsz;
    const size_t pre = a > h.data_off ? 1 : 0;                 /* one byte back to see whether a line starts at a */
    size_t len = 0;
    char *buf = slurp_range(path, a - pre, (b - a) + pre + 4096, &len);   /* 4 KiB of slack to finish the last line */
    if (!buf) return 1;
    const char *p = buf + pre, *end = buf + pre + (b - a);
    if (pre && buf[0] != '\n') { while (p < buf + len && *p != '\n') ++p; if (p < buf + len) ++p; }   /* skip the partial line */
    const char *stop = end;
    if (stop > buf + len) stop = buf + len;
    /* lines starting before `end` are ours even if they finish after it */
    const char *q = stop;
    if (q > buf && q[-1] != '\n') { while (q < buf + len && *q != '\n') ++q; }
    /* This rank's byte range is tokenised ONCE, by several threads (sub-ranges cut at line ends; the ranks of a node share
     * its cores: BICG_MTX_THREADS, else the cores this process may use / ranks, at most 32), into lists that keep every
     * triplet; lists in thread order = file order of the range. The non-zero balanced cuts are computed from those lists
     * (one all-reduce of the per-row counts), then the triplets are packed by owner for the exchange. */
    tseg *segs = NULL;
    int nseg = 0;
    int on_node = np;                                   /* the ranks of THIS node share its cores */
    {   /* a collective: EVERY rank calls it, also one whose byte range holds no line start (more ranks than entry lines) */
        MPI_Comm node;
        if (MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, me, MPI_INFO_NULL, &node) == MPI_SUCCESS) {
            MPI_Comm_size(node, &on_node);
            MPI_Comm_free(&node);
        }
    }
    const unsigned long banner_nz = h.nz;
    long nt = 1;
    if (p < q) {
        nt = loader_threads_shared((size_t)(q - p), on_node);
        h.nz = (unsigned long)-1;                       /* a byte range has no entry count of its own */
        rc = parse_threaded(p, q, &h, 0u, (unsigned)h.m, nt, &segs, &nseg);
        h.nz = banner_nz;
    } else {
        h.emitted = 0; h.lines = 0;
    }
    {   /* A malformed range must stop every rank, not leave the others in the collectives below; so must a file with FEWER
         * entry lines than its banner says (the reference's error, src/matrix.c:315-318). MORE lines: the reference reads the
         * first nz and never looks at the rest -- a prefix sum of the ranks' line counts finds the rank whose range holds the
         * nz-th line (it reads its range again up to that line); the ranks behind it drop theirs. That includes lines that do
         * not parse: the FIRST range that failed knows how many entries lie in front of it (every earlier range parsed) -- if
         * those are nz or more, it and everything behind it is never looked at, like in the reference; otherwise it reads its
         * range again up to the nz-th entry, and only a failure of THAT is an error. */
        int first_fail = rc ? me : np;
        MPI_Allreduce(MPI_IN_PLACE, &first_fail, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
        unsigned long long mine = me < first_fail ? h.lines : 0ull, before = 0, lines = 0;
        MPI_Allreduce(&mine, &lines, 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);      /* entries in front of the first failed range */
        if (first_fail < np && me >= first_fail) {
            for (int i = 0; i < nseg; ++i) free(segs[i].t);
            free(segs); segs = NULL; nseg = 0;
            h.emitted = 0; h.lines = 0; rc = 0;
            if (me == first_fail && lines < (unsigned long long)banner_nz) {
                h.nz = (unsigned long)((unsigned long long)banner_nz - lines);
                rc = parse_threaded(p, q, &h, 0u, (unsigned)h.m, nt, &segs, &nseg);
                h.nz = banner_nz;
                mine = rc ? 0ull : h.lines;
            }
        }
        MPI_Exscan(&mine, &before, 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
        if (me == 0) before = 0;
        MPI_Allreduce(&mine, &lines, 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
        int any = rc;
        MPI_Allreduce(MPI_IN_PLACE, &any, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
        if (!any && lines < (unsigned long long)h.nz) {
            if (me == 0) fprintf(stderr, "ERROR: reading matrix data: %llu entry lines, the banner says %lu.\n", lines, (unsigned long)h.nz);
            any = 6;
        }
        int again = 0;
        if (!any && lines > (unsigned long long)h.nz && before + mine > (unsigned long long)h.nz) {
            for (int i = 0; i < nseg; ++i) free(segs[i].t);
            free(segs); segs = NULL; nseg = 0;
            h.emitted = 0; h.lines = 0;
            if (before < (unsigned long long)h.nz) {
                h.nz = (unsigned long)((unsigned long long)banner_nz - before);
                again = parse_threaded(p, q, &h, 0u, (unsigned)h.m, nt, &segs, &nseg);   /* lines it has read once already */
                h.nz = banner_nz;
                if (again) fprintf(stderr, "ERROR: reading matrix data (second pass of rank %d).\n", me);
            }
        }
        MPI_Allreduce(MPI_IN_PLACE, &again, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);      /* every rank returns the error, none is left in a collective */
        if (!any && again) any = again;
        free(buf);
        if (any) { for (int i = 0; i < nseg; ++i) free(segs[i].t); free(segs); return any; }
    }
    const int *pcounts = NULL, *pdispls = NULL;
    if (part == BICG_PART_NNZ) {
        unsigned *cnt = (unsigned *)calloc(h.m ? h.m : 1, sizeof(unsigned));
        for (int sg = 0; sg < nseg; ++sg)
            for (size_t e = 0; e < segs[sg].n; ++e) cnt[segs[sg].t[e].r]++;
        MPI_Allreduce(MPI_IN_PLACE, cnt, (int)h.m, MPI_UNSIGNED, MPI_SUM, MPI_COMM_WORLD);
        bicg_partition_nnz(cnt, (unsigned)h.m, np, info->recvcounts, info->displs);
        free(cnt);
        pcounts = info->recvcounts; pdispls = info->displs;
    }
    {   /* see the serial loader: nz = entries actually emitted, over all byte ranges */
        unsigned long long tot = h.emitted;
        MPI_Allreduce(MPI_IN_PLACE, &tot, 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
        info->nz = (unsigned)tot;
        if (h.symmetric && me == 0)
            fprintf(stderr, "bicg_mtx: 'symmetric' storage: mirrored to %llu entries (the reference's block loader keeps the stored triangle only)\n", tot);
    }

    /* exchange: counts, then triplets; source-rank order = file order. Counts and displacements are in
     * TRIPLETS (a 16-byte MPI datatype), so a pair of ranks may exchange up to INT_MAX of them; beyond that
     * (> 2^31 entries to or from one rank) the int-based MPI interface cannot express the exchange at all */
    int *scnt = (int *)malloc(sizeof(int) * np), *sdsp = (int *)malloc(sizeof(int) * np);
    int *rcnt = (int *)malloc(sizeof(int) * np), *rdsp = (int *)malloc(sizeof(int) * np);
    size_t *to = (size_t *)calloc((size_t)np, sizeof(size_t));
    for (int sg = 0; sg < nseg; ++sg)
        for (size_t e = 0; e < segs[sg].n; ++e) {
            const unsigned long r = segs[sg].t[e].r;
            to[pcounts ? owner_in(r, pcounts, pdispls, np) : owner_of(r, h.m, np)]++;
        }
    size_t stot = 0;
    int too_big = 0;
    for (int r = 0; r < np; ++r) {
        if (to[r] > (size_t)INT_MAX || stot > (size_t)INT_MAX) too_big = 1;
        scnt[r] = (int)to[r]; sdsp[r] = (int)stot; stot += to[r];
    }
    MPI_Alltoall(scnt, 1, MPI_INT, rcnt, 1, MPI_INT, MPI_COMM_WORLD);
    size_t rtot = 0;
    for (int r = 0; r < np; ++r) { if (rtot > (size_t)INT_MAX) too_big = 1; rdsp[r] = (int)rtot; rtot += (size_t)rcnt[r]; }
    MPI_Allreduce(MPI_IN_PLACE, &too_big, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    if (too_big) {
        if (me == 0) fprintf(stderr, "ERROR: bicg_mtx: more than 2^31 entries to or from one rank: use more ranks or the serial loader\n");
        for (int i = 0; i < nseg; ++i) free(segs[i].t);
        free(segs); free(to); free(scnt); free(sdsp); free(rcnt); free(rdsp);
        return 7;
    }
    MPI_Datatype trip;
    MPI_Type_contiguous((int)sizeof(triplet), MPI_BYTE, &trip);
    MPI_Type_commit(&trip);
    triplet *sbuf = (triplet *)malloc(sizeof(triplet) * (stot ? stot : 1)), *rbuf = (triplet *)malloc(sizeof(triplet) * (rtot ? rtot : 1));
    for (int r = 0; r < np; ++r) to[r] = (size_t)sdsp[r];              /* cursors: lists walked in order = file order per owner */
    for (int sg = 0; sg < nseg; ++sg) {
        for (size_t e = 0; e < segs[sg].n; ++e) {
            const unsigned long r = segs[sg].t[e].r;
            sbuf[to[pcounts ? owner_in(r, pcounts, pdispls, np) : owner_of(r, h.m, np)]++] = segs[sg].t[e];
        }
        free(segs[sg].t);
    }
    free(segs); free(to);
    MPI_Alltoallv(sbuf, scnt, sdsp, trip, rbuf, rcnt, rdsp, trip, MPI_COMM_WORLD);
    MPI_Type_free(&trip);
    free(sbuf);
    const unsigned lo = (unsigned)info->displs[me], hi = lo + (unsigned)info->recvcounts[me];
    const tseg all = {rbuf, rtot};
    build_blocks(&all, 1, lo, hi, (unsigned)h.n, diag, offd);
    free(rbuf); free(scnt); free(sdsp); free(rcnt); free(rdsp);
    return 0;
}
#endif

void bicg_mtx_free(CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
{
    free(diag->val); free(diag->col); free(diag->ptr);
    free(offd->val); free(offd->col); free(offd->ptr);
    free(info->recvcounts); free(info->displs);
}
