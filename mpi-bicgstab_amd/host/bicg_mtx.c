/* bicg_mtx.c -- see bicg_mtx.h */
#include "bicg_mtx.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned r, c; double v; } triplet;

static char *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (!buf || fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
    fclose(f);
    buf[sz] = 0;
    *len = (size_t)sz;
    return buf;
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
    for (size_t e = 0; e < nt; ++e) {        /* file order inside every row */
        unsigned k = cur[t[e].r]++;
        A->col[k] = t[e].c; A->val[k] = t[e].v;
    }
    free(cur);
}

int bicg_mtx_load_block(const char *path, int rank, int nranks, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
{
    size_t len = 0;
    char *buf = slurp(path, &len);
    if (!buf) { fprintf(stderr, "ERROR: can't open file \"%s\"\n", path); return 1; }
    char *p = buf;
    if (strncmp(p, "%%MatrixMarket", 14) != 0) { fprintf(stderr, "ERROR: Could not process Matrix Market banner.\n"); free(buf); return 2; }
    char *eol = strchr(p, '\n');
    if (!eol) { free(buf); return 2; }
    *eol = 0;
    char banner[1100];
    size_t bl = strlen(p);
    for (size_t i = 0; i <= bl && i < sizeof banner - 1; ++i) banner[i] = (char)tolower((unsigned char)p[i]);
    banner[sizeof banner - 1] = 0;
    const int pattern = strstr(banner, "pattern") != NULL, integer = strstr(banner, "integer") != NULL;
    const int symmetric = strstr(banner, "symmetric") != NULL && strstr(banner, "skew") == NULL;
    if (!strstr(banner, "coordinate") || strstr(banner, "complex")) {
        fprintf(stderr, "Sorry, this application does not support Market Market type: [%s]\n", banner);
        free(buf); return 3;
    }
    p = eol + 1;
    while (*p == '%') { p = strchr(p, '\n'); if (!p) { free(buf); return 4; } ++p; }
    char *q;
    unsigned long m = strtoul(p, &q, 10); p = q;
    unsigned long n = strtoul(p, &q, 10); p = q;
    unsigned long nz = strtoul(p, &q, 10); p = q;
    if (!m || !n) { fprintf(stderr, "ERROR: Could not read matrix size.\n"); free(buf); return 5; }

    info->rows = (unsigned)m; info->cols = (unsigned)n; info->nz = (unsigned)nz;
    memcpy(info->code, pattern ? "MCPG" : (integer ? "MCIG" : "MCRG"), 4);
    if (symmetric) info->code[3] = 'S';
    info->recvcounts = (int *)malloc(sizeof(int) * (size_t)nranks);
    info->displs = (int *)malloc(sizeof(int) * (size_t)nranks);
    bicg_partition((unsigned)m, nranks, info->recvcounts, info->displs);
    const unsigned lo = (unsigned)info->displs[rank], hi = lo + (unsigned)info->recvcounts[rank];

    size_t cap = (size_t)(nz / (unsigned long)nranks) * (symmetric ? 3 : 2) + 1024, nd = 0, no = 0;
    triplet *td = (triplet *)malloc(sizeof(triplet) * cap), *to = (triplet *)malloc(sizeof(triplet) * cap);
    size_t capd = cap, capo = cap;
    for (unsigned long e = 0; e < nz; ++e) {
        unsigned long i = strtoul(p, &q, 10);
        if (q == p) { fprintf(stderr, "ERROR: reading matrix data.\n"); free(buf); return 6; }
        p = q;
        unsigned long j = strtoul(p, &q, 10); p = q;
        double v = 1.0;
        if (!pattern) { v = strtod(p, &q); p = q; }
        --i; --j;                                           /* 0-based, src/matrix.c:333-334 */
        for (int pass = 0; pass < (symmetric && i != j ? 2 : 1); ++pass) {
            unsigned long r = pass ? j : i, c = pass ? i : j;
            if (r < lo || r >= hi) continue;
            if (c >= lo && c < hi) {
                if (nd == capd) { capd *= 2; td = (triplet *)realloc(td, sizeof(triplet) * capd); }
                td[nd].r = (unsigned)(r - lo); td[nd].c = (unsigned)(c - lo); td[nd].v = v; ++nd;
            } else {
                if (no == capo) { capo *= 2; to = (triplet *)realloc(to, sizeof(triplet) * capo); }
                to[no].r = (unsigned)(r - lo); to[no].c = (unsigned)c; to[no].v = v; ++no;
            }
        }
    }
    free(buf);
    csr_from_triplets(td, nd, hi - lo, hi - lo, diag);      /* cols = local rows, src/matrix.c:343-345 */
    csr_from_triplets(to, no, hi - lo, (unsigned)n, offd);  /* cols = n,          src/matrix.c:350-352 */
    free(td); free(to);
    return 0;
}

void bicg_mtx_free(CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info)
{
    free(diag->val); free(diag->col); free(diag->ptr);
    free(offd->val); free(offd->col); free(offd->ptr);
    free(info->recvcounts); free(info->displs);
}
