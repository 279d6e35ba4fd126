/*
 * bicg_main.c -- the C host of the MI355X BiCGStab path: same command line and stdout as the
 * reference's driver (reference src/main.c), calling the solver.h entry points that
 * libbicgstab_hip.so exports.
 *
 *   mpiexec -n P bicg_solver_host <matrix.mtx> <method> [krr nrr]      (one rank per GPU)
 *   bicg_solver_host <matrix.mtx> bicgstab                             (single rank, no MPI needed)
 *
 * Set-up as in reference src/main.c:81-117: load this rank's blocks, b = A*1 (here through the
 * library's SpMV instead of MPI_csr_spmv_ovlap), x0 = 0, dispatch on the method name (:122-141).
 * Optional 5th/6th argument "--dump <prefix>" writes <prefix>.rank<p>.bin like oracle/ref_dump.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef BICG_HAVE_MPI
#include <mpi.h>
#endif

#include "bicg_mtx.h"

static double wall(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
    int np = 1, me = 0;
#ifdef BICG_HAVE_MPI
    MPI_Init(&argc, &argv);
    MPI_Comm_size(MPI_COMM_WORLD, &np);
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
#endif
    const char *dump = NULL;
    for (int i = 1; i + 1 < argc; ++i)
        if (strcmp(argv[i], "--dump") == 0) { dump = argv[i + 1]; argc = i; break; }
    if (argc < 3) {
        if (me == 0) {
            printf("Usage: %s <matrix file> <method> [options]\n", argv[0]);
            printf("Methods:\n  bicgstab\n  ca_bicgstab\n  pipe_bicgstab\n  pipe_bicgstab_rr <r> <s>\n");
        }
#ifdef BICG_HAVE_MPI
        MPI_Finalize();
#endif
        return 1;
    }
    const char *method = argv[2];
    if (me == 0) printf("Node: %d, Proc: %d\n", 1, np);   /* single node: one process per GPU */

    CSR_Matrix diag, offd;
    INFO_Matrix info;
    double t0 = wall();
    /* BICG_PARTITION=nnz: non-zero balanced row blocks instead of the reference's equal rows.
     * BICG_MTX_CACHE=<dir>: keep / reuse a binary copy of this rank's parsed blocks in <dir>. */
    const char *pm = getenv("BICG_PARTITION"), *cdir = getenv("BICG_MTX_CACHE");
    const int part = (pm && strcmp(pm, "nnz") == 0) ? BICG_PART_NNZ : BICG_PART_ROWS;
    char cpath[4096];
    int hit = 0;
    if (cdir && *cdir) {
        const char *base = strrchr(argv[1], '/');
        snprintf(cpath, sizeof cpath, "%s/%s.P%d.r%d.%s.bicgblk", cdir, base ? base + 1 : argv[1], np, me,
                 part == BICG_PART_NNZ ? "nnz" : "rows");
        hit = bicg_mtx_cache_load(cpath, argv[1], me, np, part, &diag, &offd, &info) == 0;
#ifdef BICG_HAVE_MPI
        int all = hit;                      /* the parse below is collective: all ranks or none */
        MPI_Allreduce(&hit, &all, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
        if (hit && !all) bicg_mtx_free(&diag, &offd, &info);
        hit = all;
#endif
    }
    const char *ing = getenv("BICG_INGEST");                 /* device: COO -> CSR on the GPU */
    if (ing && strcmp(ing, "device") == 0) bicg_mtx_set_block_builder(bicg_coo_to_blocks_device);
    if (!hit) {
#ifdef BICG_HAVE_MPI
        /* every rank tokenises 1/P of the file, triplets are exchanged (the reference has every rank
         * fscanf the whole file twice, src/matrix.c:315-341, 357-393) */
        if ((np > 1 ? bicg_mtx_load_block_mpi_part(argv[1], part, &diag, &offd, &info)
                    : bicg_mtx_load_block_part(argv[1], me, np, part, &diag, &offd, &info)) != 0) exit(EXIT_FAILURE);
#else
        if (bicg_mtx_load_block_part(argv[1], me, np, part, &diag, &offd, &info) != 0) exit(EXIT_FAILURE);
#endif
        if (cdir && *cdir && bicg_mtx_cache_save(cpath, argv[1], me, np, part, &diag, &offd, &info) != 0 && me == 0)
            fprintf(stderr, "bicg_solver_host: could not write the block cache %s\n", cpath);
    }
    if (me == 0 && cdir && *cdir) printf("Block cache  : %s\n", hit ? "hit" : "miss (written)");
    if (me == 0) printf("IO time      : %e [sec.]\n", wall() - t0);
    if (info.cols != info.rows) { printf("Error: matrix is not square.\n"); exit(1); }

    const unsigned nl = diag.rows;
    double *x = (double *)malloc(sizeof(double) * nl), *r = (double *)malloc(sizeof(double) * nl);
    for (unsigned i = 0; i < nl; ++i) x[i] = 1.0;          /* exact solution: all ones */
    /* the context the solver call below will reuse: plan + upload happen once */
    const double t_setup = wall();
    bicg_ctx *ctx = bicg_dropin_context(&diag, &offd, &info);
    if (!ctx) exit(EXIT_FAILURE);
    if (me == 0) printf("Setup time   : %e [sec.] (SpMV plan + upload, once per matrix)\n", wall() - t_setup);
    bicg_spmv(ctx, x, r);                                  /* b = A * 1 */
    for (unsigned i = 0; i < nl; ++i) x[i] = 0.0;          /* x0 = 0 */

    int k;
    if (strcmp(method, "bicgstab") == 0) k = bicgstab(&diag, &offd, &info, x, r);
    else if (strcmp(method, "ca_bicgstab") == 0) k = ca_bicgstab(&diag, &offd, &info, x, r);
    else if (strcmp(method, "pipe_bicgstab") == 0) k = pipe_bicgstab(&diag, &offd, &info, x, r);
    else if (strcmp(method, "pipe_bicgstab_rr") == 0) {
        if (argc != 5) {
            if (me == 0) printf("Usage for pipe_bicgstab_rr: %s <matrix file> pipe_bicgstab_rr <r> <s>\n", argv[0]);
            exit(1);
        }
        k = pipe_bicgstab_rr(&diag, &offd, &info, x, r, atoi(argv[3]), atoi(argv[4]));
    } else {
        if (me == 0) printf("Unknown method: %s\n", method);
        exit(1);
    }

    if (dump) {
        char path[4096];
        snprintf(path, sizeof path, "%s.rank%d.bin", dump, me);
        FILE *f = fopen(path, "wb");
        int nli = (int)nl;
        if (f) { fwrite(&k, sizeof(int), 1, f); fwrite(&nli, sizeof(int), 1, f); fwrite(x, 8, nl, f); fwrite(r, 8, nl, f); fclose(f); }
    }
    free(x); free(r);
    bicg_mtx_free(&diag, &offd, &info);
    bicg_comm_finalize();
#ifdef BICG_HAVE_MPI
    MPI_Finalize();
#endif
    return 0;
}
