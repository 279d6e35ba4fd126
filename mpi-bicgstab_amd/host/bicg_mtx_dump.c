/*
 * bicg_mtx_dump.c -- test driver of the Matrix-Market block loader (no GPU, no HIP library):
 *   [mpiexec -n P] bicg_mtx_dump <matrix.mtx> <out_prefix> [serial|mpi] [rows|nnz] [cache_dir]
 * writes <out_prefix>.rank<p>.bin = u32 rows, u32 ncols_offd, u32 nnz_d, u32 nnz_o, then the diag
 * ptr/col/val and offd ptr/col/val arrays, then i32 displs[P], counts[P]; tests/test_host_loader.py
 * compares them with the reference's partition and diag/offd split (reference src/matrix.c:295-308,
 * 336-392). With cache_dir the blocks make a round trip through the binary block cache first
 * (bicg_mtx_cache_save / _load) and the exit code is 7 if that does not give a valid hit.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef BICG_HAVE_MPI
#include <mpi.h>
#endif
#include "bicg_mtx.h"

int main(int argc, char **argv)
{
    int np = 1, me = 0;
#ifdef BICG_HAVE_MPI
    MPI_Init(&argc, &argv);
    MPI_Comm_size(MPI_COMM_WORLD, &np);
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
#endif
    if (argc < 3) { fprintf(stderr, "usage: %s <mtx> <out_prefix> [serial|mpi] [rows|nnz] [cache_dir]\n", argv[0]); return 2; }
    CSR_Matrix d, o;
    INFO_Matrix info;
    int rc;
    const int part = (argc > 4 && strcmp(argv[4], "nnz") == 0) ? BICG_PART_NNZ : BICG_PART_ROWS;
#ifdef BICG_HAVE_MPI
    if (argc > 3 && strcmp(argv[3], "mpi") == 0) rc = bicg_mtx_load_block_mpi_part(argv[1], part, &d, &o, &info);
    else
#endif
        rc = bicg_mtx_load_block_part(argv[1], me, np, part, &d, &o, &info);
    if (rc) return rc;
    if (argc > 5) {
        char cpath[4096];
        snprintf(cpath, sizeof cpath, "%s/blocks.P%d.r%d.bicgblk", argv[5], np, me);
        if (bicg_mtx_cache_save(cpath, argv[1], me, np, part, &d, &o, &info) != 0) return 7;
        bicg_mtx_free(&d, &o, &info);
        if (bicg_mtx_cache_load(cpath, argv[1], me, np, part, &d, &o, &info) != 0) return 7;
        /* wrong rank / partition / rank count must all miss */
        CSR_Matrix d2, o2;
        INFO_Matrix i2;
        if (bicg_mtx_cache_load(cpath, argv[1], me, np + 1, part, &d2, &o2, &i2) == 0) return 7;
        if (bicg_mtx_cache_load(cpath, argv[1], me, np, 1 - part, &d2, &o2, &i2) == 0) return 7;
    }
    char path[4096];
    snprintf(path, sizeof path, "%s.rank%d.bin", argv[2], me);
    FILE *f = fopen(path, "wb");
    unsigned hdr[4] = {d.rows, o.cols, d.ptr[d.rows], o.ptr[o.rows]};
    fwrite(hdr, 4, 4, f);
    fwrite(d.ptr, 4, d.rows + 1, f); fwrite(d.col, 4, hdr[2], f); fwrite(d.val, 8, hdr[2], f);
    fwrite(o.ptr, 4, o.rows + 1, f); fwrite(o.col, 4, hdr[3], f); fwrite(o.val, 8, hdr[3], f);
    fwrite(info.displs, 4, (size_t)np, f); fwrite(info.recvcounts, 4, (size_t)np, f);
    fclose(f);
    bicg_mtx_free(&d, &o, &info);
#ifdef BICG_HAVE_MPI
    MPI_Finalize();
#endif
    return 0;
}
