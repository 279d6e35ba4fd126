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
