/*
 * ref_dump_main.c -- ORACLE driver (test infrastructure, NOT product code).
 *
 * A small SPMD main() of our own that links the REAL reference objects (matrix.c, solver.c,
 * vector.c, mmio.c compiled from /root/reference/src by oracle/Makefile) and, unlike the
 * reference's main.c, writes the per-rank solution and residual blocks to disk so that tests can
 * compare them with the restatement (oracle/bicg_oracle.c) and with the HIP path.
 *
 *   mpiexec -n P ref_dump <matrix.mtx> <method> <out_prefix> [krr nrr]
 *
 * Set-up follows reference src/main.c:81-117: load blocks, b = A*1, x0 = 0.
 * Output: <out_prefix>.rank<p>.bin = int32 k, int32 n_loc, double x[n_loc], double r[n_loc].
 * With method "spmv" it writes y = A*(1 + 0.001*global_row) instead (k = 0, r unused = 0);
 * with method "rhs" it writes x = 0 and r = b = A*1 without solving.
 */
#include "solver.h"

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int np, me;
    MPI_Comm_size(MPI_COMM_WORLD, &np);
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
    if (argc < 4) {
        if (me == 0) fprintf(stderr, "usage: %s <mtx> <method> <out_prefix> [krr nrr]\n", argv[0]);
        MPI_Finalize();
        return 2;
    }
    INFO_Matrix info;
    info.recvcounts = (int *)malloc(sizeof(int) * np);
    info.displs = (int *)malloc(sizeof(int) * np);
    CSR_Matrix diag, offd;
    csr_init_matrix(&diag);
    csr_init_matrix(&offd);
    MPI_csr_load_matrix_block(argv[1], &diag, &offd, &info);

    int nl = (int)diag.rows, n = (int)info.rows, k = 0;
    double *x = (double *)malloc(sizeof(double) * nl), *r = (double *)malloc(sizeof(double) * nl);
    double *full = (double *)malloc(sizeof(double) * n);
    const char *method = argv[2];

    if (strcmp(method, "spmv") == 0) {
        for (int i = 0; i < nl; ++i) { x[i] = 1.0 + 0.001 * (double)(info.displs[me] + i); r[i] = 0.0; }
        double *y = (double *)malloc(sizeof(double) * nl);
        MPI_csr_spmv_ovlap(&diag, &offd, &info, x, full, y);
        memcpy(x, y, sizeof(double) * nl);
        free(y);
    } else {
        for (int i = 0; i < nl; ++i) x[i] = 1.0;
        MPI_csr_spmv_ovlap(&diag, &offd, &info, x, full, r);
        for (int i = 0; i < nl; ++i) x[i] = 0.0;
        if      (strcmp(method, "rhs") == 0)           k = 0;
        else if (strcmp(method, "bicgstab") == 0)      k = bicgstab(&diag, &offd, &info, x, r);
        else if (strcmp(method, "ca_bicgstab") == 0)   k = ca_bicgstab(&diag, &offd, &info, x, r);
        else if (strcmp(method, "pipe_bicgstab") == 0) k = pipe_bicgstab(&diag, &offd, &info, x, r);
        else if (strcmp(method, "pipe_bicgstab_rr") == 0 && argc >= 6)
            k = pipe_bicgstab_rr(&diag, &offd, &info, x, r, atoi(argv[4]), atoi(argv[5]));
        else { if (me == 0) fprintf(stderr, "unknown method %s\n", method); MPI_Finalize(); return 2; }
    }

    char path[4096];
    snprintf(path, sizeof path, "%s.rank%d.bin", argv[3], me);
    FILE *f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); MPI_Abort(MPI_COMM_WORLD, 1); }
    fwrite(&k, sizeof(int), 1, f);
    fwrite(&nl, sizeof(int), 1, f);
    fwrite(x, sizeof(double), nl, f);
    fwrite(r, sizeof(double), nl, f);
    fclose(f);
    MPI_Finalize();
    return 0;
}
