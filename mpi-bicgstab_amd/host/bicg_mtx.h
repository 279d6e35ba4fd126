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

/* Row partitions. BICG_PART_ROWS: the reference's equal-rows blocks (src/matrix.c:295-308).
 * BICG_PART_NNZ: contiguous blocks with equal non-zero counts (the reference's abandoned DYNAMIC_ROWS
 * idea, archive/matrix.c:407-446) -- the solver takes any contiguous partition through
 * INFO_Matrix.recvcounts/displs. */
int bicg_mtx_load_block_part(const char *path, int rank, int nranks, int part, CSR_Matrix *diag, CSR_Matrix *offd,
                             INFO_Matrix *info);

/* Binary cache of one rank's parsed blocks: bicg_mtx_cache_load returns 0
 * and fills diag/offd/info when `cache_path` holds the blocks of exactly this (rank, nranks,
 * partition) and `src_path` (may be NULL: no staleness check) still has the recorded size and
 * modification time; any other value means "parse the text file". Checksummed; written atomically. */
int bicg_mtx_cache_save(const char *cache_path, const char *src_path, int rank, int nranks, int part,
                        const CSR_Matrix *diag, const CSR_Matrix *offd, const INFO_Matrix *info);
int bicg_mtx_cache_load(const char *cache_path, const char *src_path, int rank, int nranks, int part,
                        CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);

/* bicg_mtx_set_block_builder (bicgstab_hip.h): replace the host counting sort that turns a rank's triplets into
 * its two CSR blocks, e.g. by bicg_coo_to_blocks_device. */

#ifdef BICG_HAVE_MPI
int bicg_mtx_load_block_mpi_part(const char *path, int part, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);
/* Collective over MPI_COMM_WORLD: every rank tokenises only ITS 1/P byte range of the file (cut at
 * line boundaries), bins the triplets by owning rank and exchanges them with MPI_Alltoallv; triplets
 * arrive in source-rank order = file order, so rows keep the reference's stored order. Parse time
 * drops from T to ~T/P (the reference parses the whole file twice per rank). */
int bicg_mtx_load_block_mpi(const char *path, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);
#endif
void bicg_mtx_free(CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);

#endif
