// bicg_plan.cpp -- host-only planning helpers (no HIP calls; unit-tested on CPU).
//
//   bicg_partition   the reference's equal-rows partition           (reference src/matrix.c:295-308)
//   bicg_halo_plan   which entries of x an offd block really needs  (replaces the full-vector
//                    MPI_Iallgatherv of reference src/matrix.c:432 by a halo)
//   bicg_row_blocks  greedy row blocks for the row-block-stream SpMV
#include "../../include/bicgstab_hip.h"

#include <algorithm>
#include <cstdint>
#include <vector>

extern "C" void bicg_partition(unsigned int n, int nranks, int *counts, int *displs)
{
    const int base = (int)(n / (unsigned)nranks), extra = (int)(n % (unsigned)nranks);
    for (int p = 0; p < nranks; ++p) {
        counts[p] = base + (p < extra ? 1 : 0);
        displs[p] = p * base + std::min(p, extra);
    }
}

extern "C" int bicg_halo_plan(const CSR_Matrix *offd, const INFO_Matrix *info, int nranks,
                              unsigned int local_rows, unsigned int *halo_cols, int *recv_counts,
                              unsigned int *renumbered)
{
    const unsigned nz = offd->ptr ? offd->ptr[offd->rows] : 0u;
    std::vector<uint32_t> uniq(offd->col, offd->col + nz);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const int h = (int)uniq.size();
    for (int p = 0; p < nranks; ++p) recv_counts[p] = 0;
    int owner = 0;
    for (int i = 0; i < h; ++i) {
        // ascending columns => owners are visited in ascending order
        while (owner + 1 < nranks && uniq[i] >= (uint32_t)info->displs[owner] + (uint32_t)info->recvcounts[owner]) ++owner;
        recv_counts[owner]++;
        halo_cols[i] = uniq[i];
    }
    for (unsigned k = 0; k < nz; ++k) {
        const uint32_t pos = (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), offd->col[k]) - uniq.begin());
        renumbered[k] = local_rows + pos;
    }
    return h;
}

// Collective, in two steps so that every rank can size its buffers between them: each rank learns
// which of ITS rows the other ranks need (the send side of the halo exchange) by sending every
// owner first the count, then the list, of the global columns it wants from it. Both steps are
// personalised exchanges through the caller's alltoallv (counts/displacements in bytes).
extern "C" int bicg_halo_send_counts(int nranks, const int *recv_counts, bicg_alltoallv_fn a2a, void *user,
                                     int *send_counts)
{
    const int P = nranks;
    std::vector<int> one(P, (int)sizeof(int)), off(P);
    for (int p = 0; p < P; ++p) off[p] = p * (int)sizeof(int);
    a2a(recv_counts, one.data(), off.data(), send_counts, one.data(), off.data(), user);
    int total = 0;
    for (int p = 0; p < P; ++p) total += send_counts[p];
    return total;
}

extern "C" int bicg_halo_send_lists(int rank, int nranks, const INFO_Matrix *info, unsigned int local_rows,
                                    const unsigned int *halo_cols, const int *recv_counts, const int *send_counts,
                                    bicg_alltoallv_fn a2a, void *user, unsigned int *send_idx)
{
    const int P = nranks;
    std::vector<int> sb(P), sd(P), rb(P), rd(P);
    int racc = 0, sacc = 0;
    for (int p = 0; p < P; ++p) {
        sb[p] = recv_counts[p] * 4; sd[p] = racc * 4; racc += recv_counts[p];   // we send the lists we want values for
        rb[p] = send_counts[p] * 4; rd[p] = sacc * 4; sacc += send_counts[p];
    }
    uint32_t none = 0;
    a2a(racc ? (const void *)halo_cols : (const void *)&none, sb.data(), sd.data(),
        sacc ? (void *)send_idx : (void *)&none, rb.data(), rd.data(), user);
    const uint32_t lo = (uint32_t)info->displs[rank];
    for (int i = 0; i < sacc; ++i) {
        if (send_idx[i] < lo || send_idx[i] - lo >= local_rows) return -1;   // request outside our rows
        send_idx[i] -= lo;
    }
    return sacc;
}

extern "C" unsigned int bicg_row_blocks(const unsigned int *ptr, unsigned int rows, unsigned int chunk,
                                        unsigned int max_rows, unsigned int *rowblk)
{
    unsigned nblk = 0;
    unsigned r = 0;
    rowblk[0] = 0;
    while (r < rows) {
        const unsigned start = r;
        const unsigned base = ptr[r];
        // always take at least one row (a row longer than the chunk becomes a block of its own)
        ++r;
        while (r < rows && r - start < max_rows && ptr[r + 1] - base <= chunk) ++r;
        rowblk[++nblk] = r;
    }
    return nblk;
}
