/*
 * bicg_mpi_shim.c -- optional bridge to the host program's MPI through WEAK symbols.
 *
 * The reference's drivers are SPMD MPI programs that call the solver collectively after MPI_Init
 * (reference src/main.c:14, 122-136) and its solvers pick the rank up from MPI_COMM_WORLD
 * (src/solver.c:37). To stay a drop-in for that calling convention without making
 * libbicgstab_hip.so depend on an MPI library, every MPI function used here is declared weak: when
 * the host executable links MPI the symbols resolve at load time, otherwise they are NULL and the
 * library runs single-rank (or with whatever bicg_comm_init_* the caller chose).
 *
 * Compiled against the mpi.h found at build time (handles such as MPI_COMM_WORLD are
 * implementation-specific constants); built without BICG_HAVE_MPI the bridge reports "no MPI".
 */
#include <stddef.h>

#ifdef BICG_HAVE_MPI
#include <mpi.h>

#pragma weak MPI_Initialized
#pragma weak MPI_Finalized
#pragma weak MPI_Comm_rank
#pragma weak MPI_Comm_size
#pragma weak MPI_Bcast
#pragma weak MPI_Allreduce
#pragma weak MPI_Alltoallv

int bicg_mpi_active(void)
{
    int flag = 0, fin = 0;
    if (!MPI_Initialized || !MPI_Comm_rank || !MPI_Comm_size || !MPI_Bcast || !MPI_Allreduce || !MPI_Alltoallv) return 0;
    MPI_Initialized(&flag);
    if (MPI_Finalized) MPI_Finalized(&fin);
    return flag && !fin;
}

void bicg_mpi_rank_size(int *rank, int *size)
{
    MPI_Comm_rank(MPI_COMM_WORLD, rank);
    MPI_Comm_size(MPI_COMM_WORLD, size);
}

void bicg_mpi_bcast_bytes(void *buf, int n, int root) { MPI_Bcast(buf, n, MPI_BYTE, root, MPI_COMM_WORLD); }

/* one packed all-reduce instead of one MPI_Iallreduce per scalar (reference src/solver.c:79, 90, 98, ...) */
void bicg_mpi_allreduce_sum(double *buf, int n, void *user)
{
    (void)user;
    MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
}

void bicg_mpi_alltoallv_bytes(const void *send, const int *scnt, const int *sdsp, void *recv, const int *rcnt,
                              const int *rdsp, void *user)
{
    (void)user;
    MPI_Alltoallv(send, scnt, sdsp, MPI_BYTE, recv, rcnt, rdsp, MPI_BYTE, MPI_COMM_WORLD);
}

#else

int bicg_mpi_active(void) { return 0; }
void bicg_mpi_rank_size(int *rank, int *size) { *rank = 0; *size = 1; }
void bicg_mpi_bcast_bytes(void *buf, int n, int root) { (void)buf; (void)n; (void)root; }
void bicg_mpi_allreduce_sum(double *buf, int n, void *user) { (void)buf; (void)n; (void)user; }
void bicg_mpi_alltoallv_bytes(const void *send, const int *scnt, const int *sdsp, void *recv, const int *rcnt,
                              const int *rdsp, void *user)
{
    (void)send; (void)scnt; (void)sdsp; (void)recv; (void)rcnt; (void)rdsp; (void)user;
}

#endif
