/*
 * ref_dump_shifted_main.c -- ORACLE driver (test infrastructure, NOT product code).
 *
 * SPMD main() of our own on top of the REAL reference's shifted solvers, so that their results at
 * P > 1 ranks can be stored as fixtures (the reference's own drivers print timings only):
 *   built without -DSWITCHING : links reference src/shifted_solver.c            (ref_dump_shifted)
 *   built with    -DSWITCHING : links reference src/shifted_switching_solver.c  (ref_dump_switching)
 * (the two reference headers share one include guard, hence two binaries).
 *
 *   mpiexec -n P ref_dump_shifted <matrix.mtx> <function> <out_prefix> <seed> <sigma_0> ... <sigma_{m-1}>
 *
 * Set-up as reference src/test_shifted.c:95-117: b = A*1 + sigma[seed]*1 (for shifted_bicgstab: b = A*1),
 * x_set = 0. Output <out_prefix>.rank<p>.bin = int32 k, int32 n_loc, int32 nsig, double b[n_loc],
 * double x_set[nsig*n_loc] (shift-major), double r[n_loc].
 */
#ifdef SWITCHING
#include "shifted_switching_solver.h"
#else
#include "shifted_solver.h"
#endif

typedef int (*seeded_fn)(CSR_Matrix *, CSR_Matrix *, INFO_Matrix *, double *, double *, double *, int, int);

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int np, me;
    MPI_Comm_size(MPI_COMM_WORLD, &np);
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
    if (argc < 6) {
        if (me == 0) fprintf(stderr, "usage: %s <mtx> <function> <out_prefix> <seed> <sigma...>\n", argv[0]);
        MPI_Finalize();
        return 2;
    }
    const char *fn = argv[2];
    const int seed = atoi(argv[4]), nsig = argc - 5;
    double *sigma = (double *)malloc(sizeof(double) * nsig);
    for (int j = 0; j < nsig; ++j) sigma[j] = strtod(argv[5 + j], NULL);

    INFO_Matrix info;
    info.recvcounts = (int *)malloc(sizeof(int) * np);
    info.displs = (int *)malloc(sizeof(int) * np);
    CSR_Matrix diag, offd;
    csr_init_matrix(&diag);
    csr_init_matrix(&offd);
    MPI_csr_load_matrix_block(argv[1], &diag, &offd, &info);

    const int nl = (int)diag.rows, n = (int)info.rows;
    double *ones = (double *)malloc(sizeof(double) * nl), *r = (double *)malloc(sizeof(double) * nl);
    double *b = (double *)malloc(sizeof(double) * nl), *full = (double *)malloc(sizeof(double) * n);
    double *x = (double *)calloc((size_t)nl * nsig, sizeof(double));
    for (int i = 0; i < nl; ++i) ones[i] = 1.0;
    MPI_csr_spmv_ovlap(&diag, &offd, &info, ones, full, r);

    int k = -1;
    seeded_fn f = NULL;
#ifdef SWITCHING
    if      (strcmp(fn, "shifted_lopbicg") == 0)                  f = shifted_lopbicg;
    else if (strcmp(fn, "shifted_lopbicg_switching") == 0)        f = shifted_lopbicg_switching;
    else if (strcmp(fn, "shifted_lopbicg_switching_noovlp") == 0) f = shifted_lopbicg_switching_noovlp;
#else
    if      (strcmp(fn, "shifted_lopbicgstab") == 0)                f = shifted_lopbicgstab;
    else if (strcmp(fn, "shifted_lopbicgstab_v2") == 0)             f = shifted_lopbicgstab_v2;
    else if (strcmp(fn, "shifted_lopbicgstab_nooverlap") == 0)      f = shifted_lopbicgstab_nooverlap;
    else if (strcmp(fn, "shifted_pipe_lopbicgstab") == 0)           f = shifted_pipe_lopbicgstab;
    else if (strcmp(fn, "shifted_pipe_lopbicgstab_nooverlap") == 0) f = shifted_pipe_lopbicgstab_nooverlap;
    else if (strcmp(fn, "shifted_bicgstab") == 0) {
        memcpy(b, r, sizeof(double) * nl);
        k = shifted_bicgstab(&diag, &offd, &info, x, r, sigma, nsig);
    }
#endif
    if (f) {
        my_daxpy(nl, sigma[seed], ones, r);               /* b = A*1 + sigma[seed]*1 */
        memcpy(b, r, sizeof(double) * nl);
        k = f(&diag, &offd, &info, x, r, sigma, nsig, seed);
    }
    if (k < 0) { if (me == 0) fprintf(stderr, "unknown function %s\n", fn); MPI_Finalize(); return 2; }

    char path[4096];
    snprintf(path, sizeof path, "%s.rank%d.bin", argv[3], me);
    FILE *out = fopen(path, "wb");
    if (!out) { fprintf(stderr, "cannot write %s\n", path); MPI_Abort(MPI_COMM_WORLD, 1); }
    fwrite(&k, sizeof(int), 1, out);
    fwrite(&nl, sizeof(int), 1, out);
    fwrite(&nsig, sizeof(int), 1, out);
    fwrite(b, sizeof(double), nl, out);
    fwrite(x, sizeof(double), (size_t)nl * nsig, out);
    fwrite(r, sizeof(double), nl, out);
    fclose(out);
    MPI_Finalize();
    return 0;
}
