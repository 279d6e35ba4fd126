/* bicg_mtx.c -- see bicg_mtx.h */
#include "bicg_mtx.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef BICG_HAVE_MPI
#include <mpi.h>
#endif

typedef struct { unsigned r, c; double v; } triplet;

typedef struct {
    unsigned long m, n, nz;
    int pattern, integer, symmetric;
    size_t data_off;      /* byte offset of the first entry line */
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

static char *slurp_range(const char *path, size_t off, size_t len, size_t *got)
{
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
    if (!strstr(banner, "coordinate") || strstr(banner, "complex")) {
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

/* tokenise entry lines in [p, end): calls emit(row, col, val) with 0-based GLOBAL indices (src/matrix.c:333-334) */
static int parse_entries(const char *p, const char *end, const mtx_header *h, unsigned long max_entries,
                         void (*emit)(void *, unsigned long, unsigned long, double), void *ctx)
{
    char *q;
    unsigned long seen = 0;
    while (p < end && seen < max_entries) {
        while (p < end && isspace((unsigned char)*p)) ++p;
        if (p >= end) break;
        if (*p == '%') { while (p < end && *p != '\n') ++p; continue; }
        unsigned long i = strtoul(p, &q, 10);
        if (q == p) { fprintf(stderr, "ERROR: reading matrix data.\n"); return 6; }
        p = q;
        unsigned long j = strtoul(p, &q, 10); p = q;
        double v = 1.0;
        if (!h->pattern) { v = strtod(p, &q); p = q; }
        --i; --j;
        emit(ctx, i, j, v);
        if (h->symmetric && i != j) emit(ctx, j, i, v);
        ++seen;
    }
    return 0;
}

static void csr_from_triplets(const triplet *t, size_t nt, unsigned rows, unsigned cols, CSR_Matrix *A)
{
    A->rows = rows; A->cols = cols; A->nz = (unsigned)nt;
    A->ptr = (unsigned *)calloc((size_t)rows + 1, sizeof(unsigned));
    A->col = (unsigned *)malloc(sizeof(unsigned) * (nt ? nt : 1));
    A->val = (double *)malloc(sizeof(double) * (nt ? nt : 1));
    for (size_t e = 0; e < nt; ++e) A->ptr[t[e].r + 1]++;
    for (unsigned i = 0; i < rows; ++i) A->ptr[i + 1] += A->ptr[i];
    unsigned *cur = (unsigned *)malloc(sizeof(unsigned) * ((size_t)rows + 1));
    memcpy(cur, A->ptr, sizeof(unsigned) * ((size_t)rows + 1));
    for (size_t e = 0; e < nt; ++e) {        /* arrival (= file) order inside every row */
        unsigned k = cur[t[e].r]++;
        A->col[k] = t[e].c; A->val[k] = t[e].v;
    }
    free(cur);
}

static void fill_info(const mtx_header *h, int nranks, INFO_Matrix *info)
{
    info->rows = (unsigned)h->m; info->cols = (unsigned)h->n; info->nz = (unsigned)h->nz;
    memcpy(info->code, h->pattern ? "MCPG" : (h->integer ? "MCIG" : "MCRG"), 4);
    if (h->symmetric) info->code[3] = 'S';
    info->recvcounts = (int *)malloc(sizeof(int) * (size_t)nranks);
    info->displs = (int *)malloc(sizeof(int) * (size_t)nranks);
    partition((unsigned)h->m, nranks, info->recvcounts, info->displs);
}

/* split this rank's triplets (global row/col) into the diag (local columns) and offd (global columns) blocks */
static void build_blocks(const triplet *t, size_t nt, unsigned lo, unsigned hi, unsigned ncols, CSR_Matrix *diag, CSR_Matrix *offd)
{
    tvec d = {0, 0, 0}, o = {0, 0, 0};
    for (size_t e = 0; e < nt; ++e) {
        if (t[e].c >= lo && t[e].c < hi) tpush(&d, t[e].r - lo, t[e].c - lo, t[e].v);
        else tpush(&o, t[e].r - lo, t[e].c, t[e].v);
    }
    csr_from_triplets(d.t, d.n, hi - lo, hi - lo, diag);      /* cols = local rows, src/matrix.c:343-345 */
    csr_from_triplets(o.t, o.n, hi - lo, ncols, offd);        /* cols = n,          src/matrix.c:350-352 */
    free(d.t); free(o.t);
}

typedef struct { tvec mine; unsigned lo, hi; } serial_ctx;
static void emit_serial(void *c, unsigned long i, unsigned long j, double v)
{
    serial_ctx *s = (serial_ctx *)c;
    if (i >= s->lo && i < s->hi) tpush(&s->mine, (unsigned)i, (unsigned)j, v);
}

int bicg_mtx_load_block(const char *path, int rank, int nranks, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
{
    size_t len = 0;
    char *buf = slurp_range(path, 0, (size_t)-1, &len);
    if (!buf) { fprintf(stderr, "ERROR: can't open file \"%s\"\n", path); return 1; }
    mtx_header h;
    int rc = parse_header(buf, &h);
    if (rc) { free(buf); return rc; }
    fill_info(&h, nranks, info);
    serial_ctx s = {{0, 0, 0}, (unsigned)info->displs[rank], (unsigned)(info->displs[rank] + info->recvcounts[rank])};
    rc = parse_entries(buf + h.data_off, buf + len, &h, h.nz, emit_serial, &s);
    free(buf);
    if (rc) return rc;
    build_blocks(s.mine.t, s.mine.n, s.lo, s.hi, (unsigned)h.n, diag, offd);
    free(s.mine.t);
    return 0;
}

#ifdef BICG_HAVE_MPI
typedef struct { tvec *bins; unsigned long m; int np; } par_ctx;
static void emit_par(void *c, unsigned long i, unsigned long j, double v)
{
    par_ctx *p = (par_ctx *)c;
    tpush(&p->bins[owner_of(i, p->m, p->np)], (unsigned)i, (unsigned)j, v);
}

int bicg_mtx_load_block_mpi(const char *path, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
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
    par_ctx pc;
    pc.bins = (tvec *)calloc((size_t)np, sizeof(tvec)); pc.m = h.m; pc.np = np;
    if (p < q) rc = parse_entries(p, q, &h, (unsigned long)-1, emit_par, &pc);
    free(buf);
    if (rc) return rc;

    /* exchange: counts, then triplets (as bytes); source-rank order = file order */
    int *scnt = (int *)malloc(sizeof(int) * np), *sdsp = (int *)malloc(sizeof(int) * np);
    int *rcnt = (int *)malloc(sizeof(int) * np), *rdsp = (int *)malloc(sizeof(int) * np);
    size_t stot = 0;
    for (int r = 0; r < np; ++r) { scnt[r] = (int)(pc.bins[r].n * sizeof(triplet)); sdsp[r] = (int)stot; stot += (size_t)scnt[r]; }
    MPI_Alltoall(scnt, 1, MPI_INT, rcnt, 1, MPI_INT, MPI_COMM_WORLD);
    size_t rtot = 0;
    for (int r = 0; r < np; ++r) { rdsp[r] = (int)rtot; rtot += (size_t)rcnt[r]; }
    char *sbuf = (char *)malloc(stot ? stot : 1), *rbuf = (char *)malloc(rtot ? rtot : 1);
    for (int r = 0; r < np; ++r) { if (scnt[r]) memcpy(sbuf + sdsp[r], pc.bins[r].t, (size_t)scnt[r]); free(pc.bins[r].t); }
    free(pc.bins);
    MPI_Alltoallv(sbuf, scnt, sdsp, MPI_BYTE, rbuf, rcnt, rdsp, MPI_BYTE, MPI_COMM_WORLD);
    free(sbuf);
    const unsigned lo = (unsigned)info->displs[me], hi = lo + (unsigned)info->recvcounts[me];
    build_blocks((const triplet *)rbuf, rtot / sizeof(triplet), lo, hi, (unsigned)h.n, diag, offd);
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
