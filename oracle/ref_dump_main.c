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
 * with method "rhs" it writes x = 0 and r = b = A*1 without solving;
 * with method "blocks" it writes what MPI_csr_load_matrix_block produced (src/matrix.c:402-419) in the layout of
 * mpi-bicgstab_amd/host/bicg_mtx_dump: uint32 {local rows, global cols, nnz diag, nnz offd}, diag ptr / col / val,
 * offd ptr / col / val, int32 displs[P], recvcounts[P] -- the pin of the C host's loader (tests/test_host_loader.py).
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

    if (strcmp(method, "blocks") == 0) {
        char bpath[4096];
        snprintf(bpath, sizeof bpath, "%s.rank%d.bin", argv[3], me);
        FILE *bf = fopen(bpath, "wb");
        if (!bf) { fprintf(stderr, "cannot write %s\n", bpath); MPI_Abort(MPI_COMM_WORLD, 1); }
        unsigned hdr[4] = {diag.rows, offd.cols, diag.ptr[diag.rows], offd.ptr[offd.rows]};
        fwrite(hdr, 4, 4, bf);
        fwrite(diag.ptr, 4, diag.rows + 1, bf); fwrite(diag.col, 4, hdr[2], bf); fwrite(diag.val, 8, hdr[2], bf);
        fwrite(offd.ptr, 4, offd.rows + 1, bf); fwrite(offd.col, 4, hdr[3], bf); fwrite(offd.val, 8, hdr[3], bf);
        fwrite(info.displs, 4, (size_t)np, bf); fwrite(info.recvcounts, 4, (size_t)np, bf);
        fclose(bf);
        MPI_Finalize();
        return 0;
    }
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
