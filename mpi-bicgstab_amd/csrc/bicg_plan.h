// bicg_plan.h -- host-only plans shared by bicg_plan.cpp and bicg_create.cpp (no HIP types).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/bicgstab_hip.h"

namespace bicg {

// plan of the persistent iteration (bicg_persist.hip): see persist_plan_host in bicg_plan.cpp
struct PersistPlan {
    uint32_t nrows = 0, nslices = 0, spw = 0, nwg = 0;     // spw: slices per workgroup = nrw * rpt
    uint32_t nrw = 0, rpt = 1;                             // row wavefronts per workgroup, rows per thread
    uint32_t win_slots = 0, max_runs = 0, max_entries = 0;
    std::vector<unsigned short> rlen, rdiag;      // [nrows] entries of a row / of its diag part
    std::vector<uint32_t> wptr;                   // [nwg + 1] runs of workgroup g
    std::vector<uint32_t> runs;                   // pairs {first column, (first slot << 16) | length}
    std::vector<uint32_t> pbase;                  // [nslices + 1]
    std::vector<double> pval;                     // padded slices
    std::vector<unsigned short> pslot;
};
bool persist_plan_host(const CSR_Matrix *diag, const unsigned *optr, const unsigned *ocol, const double *oval, unsigned gmax,
                       PersistPlan &P);

}  // namespace bicg
