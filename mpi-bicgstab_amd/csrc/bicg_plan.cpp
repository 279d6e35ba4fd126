// bicg_plan.cpp -- host-only planning helpers (no HIP calls; unit-tested on CPU).
//
//   bicg_partition   the reference's equal-rows partition           (reference src/matrix.c:295-308)
//   bicg_halo_plan   which entries of x an offd block really needs  (replaces the full-vector
//                    MPI_Iallgatherv of reference src/matrix.c:432 by a halo)
//   bicg_row_blocks  greedy row blocks for the row-block-stream SpMV
#include "../../include/bicgstab_hip.h"
#include "bicg_plan.h"
#include "bicg_parallel.h"

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


// ---- x windows (SellDev::win_*, bicg_device.h): for every marked group of `group_rows` rows, the columns its rows
// touch, merged into runs of consecutive columns; gaps of up to `gap` unused values are copied along (cheaper than
// another run). A run is {first column, (first slot << 16) | length}. Returns the number of runs, or -1 when some
// group needs more than max_slots (<= 65535) slots. runs == NULL: count only. Host-only, O(nnz + column span / 64).
// (groups are planned independently, on several threads; their runs are put together in group order afterwards)
extern "C" long bicg_window_plan(const unsigned int *ptr, const unsigned int *col, unsigned int rows, unsigned int group_rows,
                                 const char *group_mask, unsigned int max_slots, unsigned int gap, unsigned int *win_ptr,
                                 unsigned int *runs, unsigned int *slots_used)
{
    const unsigned ngroups = (rows + group_rows - 1) / group_rows;
    struct Part { std::vector<unsigned> runs; std::vector<unsigned> count; unsigned most = 0; bool fail = false; size_t g0 = 0; };
    std::vector<Part> parts((size_t)bicg::plan_threads());
    const int np = bicg::parallel_ranges(ngroups, 64, [&](size_t ga, size_t gb, int part) {
        Part &P = parts[(size_t)part];
        P.g0 = ga; P.count.assign(gb - ga, 0u);
        std::vector<unsigned long long> bm;
        std::vector<unsigned> cols;
        for (size_t g = ga; g < gb && !P.fail; ++g) {
            if (group_mask && !group_mask[g]) continue;
            const unsigned r0 = (unsigned)g * group_rows, r1 = std::min(rows, r0 + group_rows);
            const unsigned j0 = ptr[r0], j1 = ptr[r1];
            if (j0 == j1) continue;
            unsigned lo = 0xFFFFFFFFu, hi = 0;
            for (unsigned j = j0; j < j1; ++j) { lo = std::min(lo, col[j]); hi = std::max(hi, col[j]); }
            cols.clear();
            if ((unsigned long long)hi - lo < (1ull << 24)) {       // a bitmap over the group's column span
                bm.assign(((size_t)hi - lo) / 64 + 1, 0ull);
                for (unsigned j = j0; j < j1; ++j) { const unsigned d = col[j] - lo; bm[d >> 6] |= 1ull << (d & 63); }
                for (size_t w = 0; w < bm.size(); ++w)
                    for (unsigned long long bits = bm[w]; bits; bits &= bits - 1)
                        cols.push_back(lo + (unsigned)(w * 64) + (unsigned)__builtin_ctzll(bits));
            } else {
                cols.assign(col + j0, col + j1);
                std::sort(cols.begin(), cols.end());
                cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            }
            unsigned slots = 0, n = 0;
            for (size_t i = 0; i < cols.size();) {
                size_t k = i;
                while (k + 1 < cols.size() && cols[k + 1] - cols[k] <= gap + 1 && cols[k + 1] - cols[i] < 65535u) ++k;
                const unsigned len = cols[k] - cols[i] + 1;
                if (slots + len > max_slots || slots + len > 65535u) { P.fail = true; break; }
                if (runs) { P.runs.push_back(cols[i]); P.runs.push_back((slots << 16) | len); }
                ++n;
                slots += len;
                i = k + 1;
            }
            P.count[g - ga] = n;
            P.most = std::max(P.most, slots);
        }
    });
    long nruns = 0;
    unsigned most = 0;
    if (win_ptr) win_ptr[0] = 0;
    for (int p = 0; p < np; ++p) {
        const Part &P = parts[(size_t)p];
        if (P.fail) return -1;
        if (runs && !P.runs.empty()) std::copy(P.runs.begin(), P.runs.end(), runs + 2 * nruns);
        for (size_t i = 0; i < P.count.size(); ++i) {
            nruns += P.count[i];
            if (win_ptr) win_ptr[P.g0 + i + 1] = (unsigned)nruns;
        }
        most = std::max(most, P.most);
    }
    if (slots_used) *slots_used = most;
    return nruns;
}

extern "C" int bicg_set_plan_threads(int n)
{
    if (n > 0) bicg::plan_threads_setting() = n;
    if (n < 0) bicg::plan_threads_setting() = 0;        // back to the automatic count
    return bicg::plan_threads();
}

// the slot of column `c` in the window whose runs are runs[2*first .. 2*end): last run whose first column is <= c
extern "C" unsigned int bicg_window_slot(const unsigned int *runs, unsigned int first, unsigned int end, unsigned int c)
{
    unsigned a = first, b = end;
    while (b - a > 1) { const unsigned m = (a + b) / 2; if (runs[2 * m] <= c) a = m; else b = m; }
    return (runs[2 * a + 1] >> 16) + (c - runs[2 * a]);
}


// ---- plan of the persistent iteration (bicg_persist.hip, struct PersistArgs): which slices a workgroup owns, its part of
// the matrix as padded slices (diag entries first, then offd entries in the x_ext numbering [local rows | halo positions])
// with window slots instead of columns, and the window runs. Host-only; false when the block does not qualify.
namespace bicg {
bool persist_plan_host(const CSR_Matrix *diag, const unsigned *optr, const unsigned *ocol, const double *oval, unsigned gmax,
                       PersistPlan &P)
{
    constexpr uint32_t kSlice = 64;
    const uint32_t nrows = diag->rows;
    const bool multi = optr != nullptr;
    if (nrows == 0 || gmax < 1) return false;
    P = PersistPlan{};
    P.nrows = nrows;
    P.nslices = (nrows + kSlice - 1) / kSlice;
    // slices per workgroup -> (row wavefronts, rows per thread): 15 row wavefronts + the communication wavefront with one or two
    // rows per thread (4 wavefronts per SIMD), or 7 + 1 wavefronts with eight rows per thread (2 per SIMD: twice the registers)
    const uint32_t per = (P.nslices + gmax - 1) / gmax;
    if (per <= 15) { P.rpt = 1; P.nrw = per; }
    else if (per <= 30) { P.rpt = 2; P.nrw = (per + 1) / 2; }
    else if (per <= 56) { P.rpt = 8; P.nrw = (per + 7) / 8; }
    else return false;
    P.spw = P.nrw * P.rpt;
    P.nwg = (P.nslices + P.spw - 1) / P.spw;
    const uint32_t grows = P.spw * kSlice, nslices = P.nslices, nwg = P.nwg, spw = P.spw;

    // merged rows in x_ext numbering
    std::vector<uint32_t> mptr(nrows + 1, 0u);
    for (uint32_t r = 0; r < nrows; ++r) mptr[r + 1] = mptr[r] + (diag->ptr[r + 1] - diag->ptr[r]) + (multi ? optr[r + 1] - optr[r] : 0u);
    std::vector<uint32_t> mcol(mptr[nrows] ? mptr[nrows] : 1);
    std::vector<double> mval(mptr[nrows] ? mptr[nrows] : 1);
    P.rlen.assign(nrows, 0); P.rdiag.assign(nrows, 0);
    std::vector<char> too_long((size_t)bicg::plan_threads(), 0);
    bicg::parallel_ranges(nrows, 4096, [&](size_t ra, size_t rb, int part) {       // (rows on several threads: each writes its own rows)
        for (uint32_t r = (uint32_t)ra; r < (uint32_t)rb; ++r) {
            uint32_t at = mptr[r];
            const uint32_t dl = diag->ptr[r + 1] - diag->ptr[r], ol = multi ? optr[r + 1] - optr[r] : 0u;
            if (dl + ol > 65535u) { too_long[(size_t)part] = 1; return; }
            for (uint32_t j = diag->ptr[r]; j < diag->ptr[r + 1]; ++j, ++at) { mcol[at] = diag->col[j]; mval[at] = diag->val[j]; }
            if (multi) for (uint32_t j = optr[r]; j < optr[r + 1]; ++j, ++at) { mcol[at] = ocol[j]; mval[at] = oval[j]; }
            P.rlen[r] = (unsigned short)(dl + ol); P.rdiag[r] = (unsigned short)dl;
        }
    });
    for (char t : too_long) if (t) return false;
    // windows: runs of consecutive columns per workgroup; a run never straddles the local / halo boundary
    const uint32_t max_slots = 16384;
    std::vector<uint32_t> wptr0(nwg + 1, 0u);
    const long nruns = bicg_window_plan(mptr.data(), mcol.data(), nrows, grows, nullptr, max_slots, 8, nullptr, nullptr, nullptr);
    if (nruns < 0) return false;
    std::vector<uint32_t> runs0(2 * (size_t)nruns + 2);
    bicg_window_plan(mptr.data(), mcol.data(), nrows, grows, nullptr, max_slots, 8, wptr0.data(), runs0.data(), &P.win_slots);
    P.wptr.assign(nwg + 1, 0u);
    for (uint32_t g = 0; g < nwg; ++g) {
        P.wptr[g] = (uint32_t)(P.runs.size() / 2);
        for (uint32_t i = wptr0[g]; i < wptr0[g + 1]; ++i) {
            const uint32_t c0 = runs0[2 * i], slot0 = runs0[2 * i + 1] >> 16, len = runs0[2 * i + 1] & 0xFFFFu;
            if (c0 < nrows && c0 + len > nrows) {
                const uint32_t l1 = nrows - c0;
                P.runs.push_back(c0); P.runs.push_back((slot0 << 16) | l1);
                P.runs.push_back(nrows); P.runs.push_back(((slot0 + l1) << 16) | (len - l1));
            } else {
                P.runs.push_back(c0); P.runs.push_back((slot0 << 16) | len);
            }
        }
        P.max_runs = std::max<uint32_t>(P.max_runs, (uint32_t)(P.runs.size() / 2) - P.wptr[g]);
    }
    P.wptr[nwg] = (uint32_t)(P.runs.size() / 2);
    if (P.max_runs > 1024) return false;
    auto slot_of = [&](uint32_t g, uint32_t col) -> uint32_t {
        uint32_t a = P.wptr[g], b = P.wptr[g + 1];
        while (b - a > 1) { const uint32_t m = (a + b) / 2; if (P.runs[2 * m] <= col) a = m; else b = m; }
        return (P.runs[2 * a + 1] >> 16) + (col - P.runs[2 * a]);
    };
    // padded slices
    P.pbase.assign(nslices + 1, 0u);
    for (uint32_t sl = 0; sl < nslices; ++sl) {
        uint32_t longest = 0;
        for (uint32_t r = sl * kSlice; r < std::min(nrows, (sl + 1) * kSlice); ++r) longest = std::max<uint32_t>(longest, P.rlen[r]);
        const uint64_t next = (uint64_t)P.pbase[sl] + (uint64_t)longest * kSlice;
        if (next >= 0xFFFFFF00ull) return false;
        P.pbase[sl + 1] = (uint32_t)next;
    }
    const size_t entries = P.pbase[nslices];
    P.pval.assign(entries ? entries : 1, 0.0);
    P.pslot.assign(entries ? entries : 1, 0);
    for (uint32_t g = 0; g < nwg; ++g) {
        const uint32_t s0 = g * spw, s1 = std::min(nslices, s0 + spw);
        P.max_entries = std::max(P.max_entries, P.pbase[s1] - P.pbase[s0]);
    }
    bicg::parallel_ranges(nwg, 4, [&](size_t ga, size_t gb, int) {                   // (a workgroup's slices are its own range of pval / pslot)
        for (uint32_t g = (uint32_t)ga; g < (uint32_t)gb; ++g) {
            const uint32_t s0 = g * spw, s1 = std::min(nslices, s0 + spw);
            for (uint32_t r = s0 * kSlice; r < std::min(nrows, s1 * kSlice); ++r) {
                const uint32_t sl = r / kSlice, lane = r % kSlice;
                for (uint32_t j = mptr[r], k = 0; j < mptr[r + 1]; ++j, ++k) {
                    const size_t e = (size_t)P.pbase[sl] + (size_t)k * kSlice + lane;
                    P.pval[e] = mval[j];
                    P.pslot[e] = (unsigned short)slot_of(g, mcol[j]);
                }
            }
        }
    });
    return true;
}
}  // namespace bicg

// C view for tests: summary = {slices per workgroup, nwg, window slots, most runs of a workgroup, most matrix entries of a
// workgroup, entries, runs, rows per thread}; the arrays (sizes known from a first call with NULL arrays) are filled when given. offd_renumbered: columns =
// rows + halo position (bicg_halo_plan), or NULL for one rank. Returns 0 when the block qualifies.
extern "C" int bicg_persist_plan(const CSR_Matrix *diag, const CSR_Matrix *offd_renumbered, unsigned int gmax, unsigned int summary[8],
                                 unsigned int *pbase, unsigned short *pslot, double *pval, unsigned short *rlen, unsigned short *rdiag,
                                 unsigned int *win_ptr, unsigned int *win_runs)
{
    bicg::PersistPlan P;
    const bool multi = offd_renumbered != nullptr;
    if (!bicg::persist_plan_host(diag, multi ? offd_renumbered->ptr : nullptr, multi ? offd_renumbered->col : nullptr,
                                 multi ? offd_renumbered->val : nullptr, gmax, P))
        return 1;
    const unsigned s[8] = {P.spw, P.nwg, P.win_slots, P.max_runs, P.max_entries, P.pbase[P.nslices], (unsigned)(P.runs.size() / 2), P.rpt};
    for (int i = 0; i < 8; ++i) summary[i] = s[i];
    if (pbase) std::copy(P.pbase.begin(), P.pbase.end(), pbase);
    if (pslot) std::copy(P.pslot.begin(), P.pslot.begin() + P.pbase[P.nslices], pslot);
    if (pval) std::copy(P.pval.begin(), P.pval.begin() + P.pbase[P.nslices], pval);
    if (rlen) std::copy(P.rlen.begin(), P.rlen.end(), rlen);
    if (rdiag) std::copy(P.rdiag.begin(), P.rdiag.end(), rdiag);
    if (win_ptr) std::copy(P.wptr.begin(), P.wptr.end(), win_ptr);
    if (win_runs) std::copy(P.runs.begin(), P.runs.end(), win_runs);
    return 0;
}
