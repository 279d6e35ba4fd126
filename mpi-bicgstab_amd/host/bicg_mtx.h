/*
 * bicg_mtx.h -- Matrix-Market block loader of the C host.
 *
 * Produces what MPI_csr_load_matrix_block produces (reference src/matrix.c:402-419): per rank the
 * diag block (local columns) and the offd block (global columns) in CSR, entries of a row in FILE
 * order (the reference's stable row sort, src/matrix.c:135-183), and the equal-rows partition of
 * src/matrix.c:295-308. Unlike the reference it reads and tokenises the file ONCE per rank
 * (the reference fscanf()s it twice, src/matrix.c:315-341 and 357-393).
 */
#ifndef BICG_MTX_H
#define BICG_MTX_H

#include "bicgstab_hip.h"

/* Returns 0 on success; on failure prints to stderr and returns non-zero.
 * Arrays inside diag/offd/info are malloc'ed; release with bicg_mtx_free.
 * Every rank reads and tokenises the whole file (no communication). */
int bicg_mtx_load_block(const char *path, int rank, int nranks, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);

#ifdef BICG_HAVE_MPI
/* Collective over MPI_COMM_WORLD: every rank tokenises only ITS 1/P byte range of the file (cut at
 * line boundaries), bins the triplets by owning rank and exchanges them with MPI_Alltoallv; triplets
 * arrive in source-rank order = file order, so rows keep the reference's stored order. Parse time
 * drops from T to ~T/P (the reference parses the whole file twice per rank). */
int bicg_mtx_load_block_mpi(const char *path, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);
#endif
void bicg_mtx_free(CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);

#endif
