// bicg_create.cpp -- building a context: the plan of the diag block (sliced-ELL layouts, lists, descriptors, the stencil and
// persistent plans), device uploads, streams, and tearing it down again. Split from bicg_solver.cpp in round 5; see bicg_host.h.
#include "bicg_host.h"

std::vector<bicg_ctx *> g_live;

bool all_ranks(Comm *comm, bool mine)
{
    const int P = comm->nranks;
    if (P == 1) return mine;
    std::vector<int> cnt(P, (int)sizeof(int)), dsp(P), out(P, mine ? 1 : 0), in(P, 0);
    for (int p = 0; p < P; ++p) dsp[p] = p * (int)sizeof(int);
    comm->alltoallv_host(out.data(), cnt.data(), dsp.data(), in.data(), cnt.data(), dsp.data());
    in[comm->rank] = mine ? 1 : 0;
    for (int p = 0; p < P; ++p) if (!in[p]) return false;
    return true;
}


static void build_stencil_plan(bicg_ctx *c, uint32_t nslices, uint32_t nrows, const std::vector<uint4> &d, const std::vector<int> &uoff,
                               const std::vector<double> &uval, const unsigned short *rmask_host);
static void build_slice_desc(bicg_ctx *c, uint32_t nslices, uint32_t nrows, const uint32_t *slice_len, const std::vector<uint32_t> &ubase,
                             const std::vector<uint32_t> &vbase, const std::vector<uint32_t> &mbase, const std::vector<int> &uoff,
                             const std::vector<double> &uval, const unsigned short *rmask_host)
{
    if (vbase.empty() || ubase.empty() || (uint64_t)nrows >= (1ull << 29)) return;
    if (plan_off("desc")) return;
    std::vector<uint4> d(nslices);
    bool all_lists = !plan_off("lists") && nrows % kGroupRows == 0;
    for (uint32_t sl = 0; sl < nslices; ++sl) {
        const uint32_t ub = ubase[sl], vb = vbase[sl], mb = mbase.empty() ? 0xFFFFFFFFu : mbase[sl];
        uint32_t len = slice_len[sl] & 0xFFFFu, kind = kSliceGeneral, w = 0;
        if (ub != 0xFFFFFFFFu && slice_len[sl] <= 0xFFFFu) {
            kind = kSliceUniform;
            if (vb != 0xFFFFFFFFu) {
                kind = kSliceConstant;
                if (mb != 0xFFFFFFFFu) { kind = kSliceMasked; len = mb >> 26; w = mb & 0x03FFFFFFu; }
            }
        }
        d[sl] = make_uint4(len | (kind << 16), kind >= kSliceConstant ? ub : 0u, kind >= kSliceConstant ? vb : 0u, w);
        if ((uint64_t)sl * kSliceRows < nrows && (kind < kSliceConstant || len == 0 || len > 8u)) all_lists = false;
    }
    if (all_lists) {                              // (SellDev::all_lists: the distances once more, as byte offsets)
        std::vector<int> u8(uoff.size());
        for (size_t i = 0; i < uoff.size(); ++i) u8[i] = (int)((uint32_t)uoff[i] * 8u);      // (modulo 2^32: the product adds it to the row's byte offset modulo 2^32)
        c->s_uoff8 = dev_upload(u8.data(), u8.size());
        c->sell_all_lists = true;
        // SellDev::ystride from the longest list (the interior's): its second-largest distance is a grid line when it is a multiple
        // of 64 rows. (Only the speed depends on the guess: any value gives every slice to exactly one wavefront.)
        uint32_t best_len = 0, best_at = 0;
        for (uint32_t sl = 0; sl < nslices; ++sl) { const uint32_t l = d[sl].x & 0xFFFFu; if ((d[sl].x >> 16) == kSliceConstant && l > best_len) { best_len = l; best_at = d[sl].y; } }
        // (measured, 512^3: 0.923 against 0.929 ms per product, 256^3 0.146 against 0.123 ms -- off unless BICG_SELL_YGROUP=1)
        if (best_len >= 5 && knob_x("BICG_SELL_YGROUP") && atoi(knob_x("BICG_SELL_YGROUP")) != 0) {
            std::vector<int> dist(uoff.begin() + best_at, uoff.begin() + best_at + best_len);
            std::sort(dist.begin(), dist.end());
            const int line = dist[best_len - 2];
            const uint32_t S = line > 0 ? (uint32_t)line / kSliceRows : 0u;
            if (S >= 1 && (uint32_t)line % kSliceRows == 0 && (S & (S - 1u)) == 0 && nslices % (4u * S) == 0) c->sell_ystride = (int)S;   // (a power of two: shifts in the kernel)
        }
    }
    c->s_desc = dev_upload(d.data(), d.size());
    c->matrix_bytes += 8ull * nslices;          // 16 bytes of descriptor per slice where base + length were counted
    if (all_lists) build_stencil_plan(c, nslices, nrows, d, uoff, uval, rmask_host);
}

// The plane-marching product (struct StencilDev, bicg_stencil.hip): is this block the 7-point stencil of a grid? Decided from the
// lists alone -- the interior's list must be (-sz, -sy, -1, 0, +1, +sy, +sz) with sy a multiple of 64 rows, sz a multiple of sy,
// the rows a multiple of sz, and every other list a sub-sequence of it in the same order. Values may differ from list to list
// (every (distance list, value list) pair gets a table entry); rows of masked slices get their entries as canonical bits.
static void build_stencil_plan(bicg_ctx *c, uint32_t nslices, uint32_t nrows, const std::vector<uint4> &d, const std::vector<int> &uoff,
                               const std::vector<double> &uval, const unsigned short *rmask_host)
{
    if (plan_off("stencil")) return;
    uint32_t best_at = 0, best_len = 0;
    // the interior's list: the longest one, of a constant slice or (a grid one x segment wide has no other) of a masked one
    for (uint32_t sl = 0; sl < nslices; ++sl) { const uint32_t l = d[sl].x & 0xFFFFu; if ((d[sl].x >> 16) >= kSliceConstant && l > best_len) { best_len = l; best_at = d[sl].y; } }
    if (best_len != 7) return;
    const int *L = uoff.data() + best_at;
    if (!(L[3] == 0 && L[2] == -1 && L[4] == 1 && L[5] > 1 && L[6] > L[5] && L[1] == -L[5] && L[0] == -L[6])) return;
    const uint32_t sy = (uint32_t)L[5], sz = (uint32_t)L[6];
    if (sy % kSliceRows || sz % sy || nrows % sz || sy / kSliceRows > 64u) return;
    const uint32_t nxs = sy / kSliceRows, ny = sz / sy, nz = nrows / sz;
    if (ny % 2u) return;
    const int canon[7] = {-(int)sz, -(int)sy, -1, 0, 1, (int)sy, (int)sz};
    struct Entry { StencilTab t; signed char pos[8]; };
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> pairs;
    std::vector<Entry> entries;
    std::vector<uint32_t> code(nslices), which(nslices);
    unsigned long long mcols = 0;
    for (uint32_t sl = 0; sl < nslices; ++sl) {
        const uint32_t kind = d[sl].x >> 16, len = d[sl].x & 0xFFFFu;
        const auto key = std::make_tuple(d[sl].y, d[sl].z, len);
        auto it = pairs.find(key);
        if (it == pairs.end()) {
            if (entries.size() >= 65536u) return;
            Entry e;
            memset(&e, 0, sizeof e);
            int cpos = -1;
            for (uint32_t k = 0; k < len; ++k) {
                int at = -1;
                for (int q = cpos + 1; q < 7; ++q) if (canon[q] == uoff[d[sl].y + k]) { at = q; break; }
                if (at < 0) return;                                   // a distance the grid does not have, or out of order: not this product
                cpos = at;
                e.t.v[at] = uval[d[sl].z + k];
                e.t.bits |= 1ull << at;
                e.pos[k] = (signed char)at;
            }
            it = pairs.emplace(key, (uint32_t)entries.size()).first;
            entries.push_back(e);
        }
        const uint32_t xs = sl % nxs, line = sl / nxs, yy = line % ny, zz = line / ny;
        which[sl] = it->second;
        code[((size_t)zz * nxs + xs) * ny + yy] = it->second;
        if (kind == kSliceMasked) mcols |= 1ull << xs;
    }
    const uint32_t nmc = (uint32_t)__builtin_popcountll(mcols);
    std::vector<unsigned char> cmask;
    if (nmc) {
        std::vector<unsigned short> rm_dl;
        if (!rmask_host) {                                            // the device plan wrote the rows' masks on the GPU
            uint32_t top = 0;
            for (uint32_t sl = 0; sl < nslices; ++sl) if ((d[sl].x >> 16) == kSliceMasked) top = std::max(top, d[sl].w + 1u);
            rm_dl.resize((size_t)top * kSliceRows);
            BICG_HIP(hipMemcpy(rm_dl.data(), c->s_rmask, sizeof(unsigned short) * rm_dl.size(), hipMemcpyDeviceToHost));
            rmask_host = rm_dl.data();
        }
        cmask.assign((size_t)(nslices / nxs) * nmc * kSliceRows, 0);
        parallel_ranges(nslices, 4096, [&](size_t s0, size_t s1, int) {
            for (size_t sl = s0; sl < s1; ++sl) {
                const uint32_t xs = (uint32_t)(sl % nxs);
                if (!((mcols >> xs) & 1ull)) continue;
                const uint32_t dense = (uint32_t)__builtin_popcountll(mcols & ((1ull << xs) - 1ull));
                unsigned char *out = cmask.data() + ((sl / nxs) * nmc + dense) * kSliceRows;
                const Entry &e = entries[which[sl]];
                if ((d[sl].x >> 16) == kSliceMasked) {
                    const unsigned short *pm = rmask_host + (size_t)d[sl].w * kSliceRows;
                    const uint32_t len = d[sl].x & 0xFFFFu;
                    for (uint32_t l = 0; l < kSliceRows; ++l) {
                        unsigned bits = 0;
                        for (uint32_t k = 0; k < len; ++k) if ((pm[l] >> k) & 1u) bits |= 1u << e.pos[k];
                        out[l] = (unsigned char)bits;
                    }
                } else {
                    for (uint32_t l = 0; l < kSliceRows; ++l) out[l] = (unsigned char)e.t.bits;
                }
            }
        });
    }
    std::vector<StencilTab> tab(entries.size());
    for (size_t i = 0; i < entries.size(); ++i) tab[i] = entries[i].t;
    // The wide form (bicg_stencil.hip, k_spmv_stencil_w): 2 or 4 rows per lane. It needs ONE value per canonical position -- every
    // table entry agrees with the full one wherever it has an entry -- and whole groups of x segments per line.
    uint32_t wide = 0, ref = 0;
    {
        bool have = false, same = true;
        for (size_t i = 0; i < tab.size() && !have; ++i) if ((tab[i].bits & 0x7Full) == 0x7Full) { ref = (uint32_t)i; have = true; }
        for (size_t i = 0; have && i < tab.size(); ++i)
            for (int k = 0; k < 7; ++k)
                if (((tab[i].bits >> k) & 1ull) && memcmp(&tab[i].v[k], &tab[ref].v[k], sizeof(double)) != 0) same = false;
        // two rows per lane wherever the grid is large enough for the traffic to matter (512^3: product 0.461 -> 0.423 ms, 256^3:
        // 0.0535 -> 0.052 ms, plain iteration 0.441 -> 0.424 ms; four rows per lane need 200-256 registers and lose what two gain:
        // profiles/r06/stencil_notes.txt)
        uint32_t want = nrows >= (1u << 22) ? 2u : 0u;
        if (const char *v = plan_tok("wide")) want = (uint32_t)atoi(v);
        if (have && same && (want == 2u || want == 4u) && nxs % want == 0) wide = want;
        else if (want && c->rank == 0 && plan_tok("wide")) fprintf(stderr, "bicgstab_hip: BICG_PLAN wide=%u not taken (%s)\n", want, !(want == 2u || want == 4u) ? "2 or 4 rows per lane" : nxs % want ? "the lines are not whole groups of segments" : "values differ between the lists of the block");
    }
    // lines per wavefront and planes per tile: enough workgroups for several rounds of the 1024 a GPU holds, tiles as deep as that allows
    uint32_t lines = 0, zl = 0;
    const uint32_t nxt = wide ? nxs / wide : nxs;          // tiles per line
    {
        static const uint32_t cand[][2] = {{4, 64}, {4, 32}, {4, 16}, {2, 32}, {2, 16}, {4, 8}, {2, 8}, {2, 4}};
        const uint64_t enough = wide ? 1000 : 3000;        // (the wide form keeps fewer, larger workgroups resident)
        uint64_t most = 0;
        for (auto &cd : cand) {
            if (ny % cd[0] || (!wide && cd[1] > 32)) continue;
            const uint64_t wgs = (uint64_t)nxt * ((ny + 4 * cd[0] - 1) / (4 * cd[0])) * ((nz + cd[1] - 1) / cd[1]);
            if (wgs >= enough) { lines = cd[0]; zl = cd[1]; break; }
            if (wgs > most) { most = wgs; lines = cd[0]; zl = cd[1]; }
        }
        if (const char *v = plan_tok("lines")) { const uint32_t r = (uint32_t)atoi(v); if ((r == 2 || r == 4) && ny % r == 0) lines = r; }
        if (const char *v = plan_tok("planes")) {
            // every workgroup of the product publishes one row of partial sums: the tiling must not need more rows than the table
            // has (ctx_state: max(256-row groups, kMaxGrid) + 64) -- thin grids with few planes per tile would
            const int z = atoi(v);
            const uint64_t wgs = z >= 1 ? (uint64_t)nxt * ((ny + 4 * lines - 1) / (4 * lines)) * ((nz + (uint32_t)z - 1) / (uint32_t)z) : 0;
            const uint64_t room = std::max<uint64_t>(((uint64_t)nrows + kGroupRows - 1) / kGroupRows, (uint64_t)kMaxGrid);
            if (z >= 1 && wgs <= room) zl = (uint32_t)z;
            else if (c->rank == 0) fprintf(stderr, "bicgstab_hip: BICG_PLAN planes=%s ignored (%llu workgroups, room for %llu partial-sum rows)\n", v, (unsigned long long)wgs, (unsigned long long)room);
        }
    }
    if (wide) {
        std::vector<uint32_t> wb((size_t)nz * nxt * ny);
        parallel_ranges(wb.size(), 65536, [&](size_t i0, size_t i1, int) {
            for (size_t i = i0; i < i1; ++i) {
                const size_t yy = i % ny, xw = (i / ny) % nxt, zz = i / ((size_t)ny * nxt);
                uint32_t word = 0;
                for (uint32_t q = 0; q < wide; ++q) word |= (uint32_t)(tab[code[(zz * nxs + xw * wide + q) * ny + yy]].bits & 0x7Full) << (8u * q);
                wb[i] = word;
            }
        });
        c->st_wbits = dev_upload(wb.data(), wb.size());
    }
    c->st_code = dev_upload(code.data(), code.size());
    c->st_tab = dev_upload(tab.data(), tab.size());
    if (nmc) c->st_cmask = dev_upload(cmask.data(), cmask.size());
    // Input + output vector far beyond the 256 MiB Infinity Cache (512^3: 2 x 1 GiB): y is stored non-temporally and the tiles go to
    // the XCDs round-robin (product 0.480 -> 0.460 ms, CA-BiCGStab 5.40 -> 5.31 ms per iteration); a grid whose vectors the cache
    // holds (256^3) keeps ordinary stores and the XCD-contiguous order (0.053 against 0.061 ms): profiles/r05/stencil_sweep_xcd_nt.txt
    const bool st_big = 16.0 * (double)nrows > 2.0 * 256.0 * 1048576.0;
    // ... and since round 6 (the wide form) in the sweep order: an XCD takes its own eighth of the line blocks plane block by plane
    // block (0.428 -> 0.423 ms; L2 misses 13.9 M -> 12.8 M per product)
    int st_xcd = knob_x("BICG_STENCIL_XCD") ? atoi(knob_x("BICG_STENCIL_XCD")) : (st_big ? (wide ? 2 : 0) : 1);
    if (st_xcd == 2 && ((ny + 4 * lines - 1) / (4 * lines)) % 8u) st_xcd = st_big ? 0 : 1;       // (the sweep order deals whole line blocks to the XCDs)
    const int st_nt = knob_x("BICG_STENCIL_NT") ? atoi(knob_x("BICG_STENCIL_NT")) : (st_big ? 1 : 0);
    c->st = StencilDev{1, sy, sz, nxs, ny, nz, 0u, nz, zl, lines, nmc, st_xcd, st_nt, mcols, c->st_code, c->st_tab, c->st_cmask, wide, ref, c->st_wbits};
    if (const char *v = plan_tok("ca-fuse")) c->ca_fuse = atoi(v) != 0;
    // what this product streams from the matrix side: 4 bytes per slice, one byte per row of the masked x segments
    // (the wide form: 4 bytes per `wide` slices)
    c->stencil_matrix_bytes = 4ull * nslices / (wide ? wide : 1u) + (uint64_t)cmask.size();
    if (getenv("BICG_PLAN_TRACE"))
        fprintf(stderr, "bicgstab_hip: plane-marching product: %u x %u x %u grid (x segments of 64 rows: %u), %zu list pairs, %u masked x segments, %u lines x %u planes per wavefront, %u rows per lane, %u workgroups\n",
                sy, ny, nz, nxs, tab.size(), nmc, lines, zl, wide ? wide : 1u, stencil_grid(c->st));
}


// Very large structured blocks (the 512^3 Laplacian: 524 288 row groups, z neighbours 262 144 rows = 2 MB of x away).
//  * groups per workgroup: with one 256-row group of 7-entry rows per workgroup the per-workgroup part of a product with dots
//    (block sum, hand-over of the partials) is a third of the kernel (2.16 ms without dots, 2.76 / 3.13 ms with one / two);
//    workgroups take ceil(groups / 65536) contiguous groups each.
//  * order of the groups: an XCD sweeps its eighth of the rows plane by plane, and the three planes a sweep front touches (6 MB
//    of x) do not fit its 4 MB L2 -- every x value comes from the Infinity Cache three times. The groups of an XCD's share are
//    therefore taken block by block through the planes: B consecutive groups of plane z, the same B of plane z + 1, ... so that
//    what a block fetched as its far neighbours is still in the L2 when it becomes the block's own rows. Only the ORDER of the
//    list changes (SpmvArgs::glist): rows, sums of a row and results are those of the natural order; the dot partials are
//    added in list order (a different, equally fixed association).
// BICG_SELL_BLOCK = B (groups, default 256; 0: natural order).
void sell_order_for_big_grids(bicg_ctx *c, uint32_t ngroups)
{
    if (!knob_x("BICG_SELL_GPW") && !knob_x("BICG_SELL_GPW_DOTS")) c->sell_gpw = c->sell_gpw_dots = (int)std::max<uint32_t>(1u, (ngroups + 65535u) / 65536u);
    if (const char *sv = knob_x("BICG_SELL_GPW")) c->sell_gpw = std::max(1, atoi(sv));
    if (const char *sv = knob_x("BICG_SELL_GPW_DOTS")) c->sell_gpw_dots = std::max(1, atoi(sv));
    const uint32_t B = knob_x("BICG_SELL_BLOCK") ? (uint32_t)atoi(knob_x("BICG_SELL_BLOCK")) : 256u;
    const uint32_t P = (c->far_rows + kGroupRows / 2) / kGroupRows;          // groups per plane
    if (B == 0 || P < 4 * B || c->sell_gpw != c->sell_gpw_dots || (uint64_t)c->far_rows * 24ull <= (3ull << 19)) return;   // three planes fit half an L2
    const uint32_t nblocks = sell_grid(ngroups, c->sell_gpw), each = (ngroups + nblocks - 1) / nblocks;
    std::vector<uint32_t> list(ngroups);
    for (uint32_t x = 0; x <= 8; ++x) {
        // XCD x's share of the list (the last segment: what the division left over); within it block y of every plane, plane
        // after plane, then block y + 1 ... -- the order a sort by (block, group) would give, enumerated directly
        const uint32_t s0 = std::min<uint64_t>(ngroups, (uint64_t)x * (nblocks / 8u) * each);
        const uint32_t s1 = x == 8 ? ngroups : std::min<uint64_t>(ngroups, (uint64_t)(x + 1) * (nblocks / 8u) * each);
        uint32_t o = s0;
        for (uint32_t y0 = 0; y0 < P && o < s1; y0 += B)
            for (uint64_t z0 = s0; z0 < s1; z0 += P)
                for (uint64_t g = z0 + y0; g < std::min<uint64_t>({(uint64_t)s1, z0 + y0 + B, z0 + P}); ++g) list[o++] = (uint32_t)g;
    }
    if (c->glist_int) BICG_HIP(hipFree(c->glist_int));
    c->glist_int = dev_upload(list.data(), list.size());
    c->glist_int_identity = false;
    c->sell_blocked = B;
}

// ---------------------------------------------------------------- persistent pipelined iteration: plan
// Which rows a workgroup owns, its part of the matrix in padded slices (diag entries first, then offd entries in the
// x_ext numbering [local rows | halo positions]) with window slots instead of columns, the window runs, and -- multi
// rank -- the send-list entries of every workgroup. Returns false when the block does not qualify.
bool persist_build(bicg_ctx *c, const CSR_Matrix *diag, const std::vector<uint32_t> &optr, const std::vector<uint32_t> &ocol,
                   const std::vector<double> &oval, const std::vector<uint32_t> &send_idx, const std::vector<unsigned long long> &dst0,
                   const std::vector<unsigned long long> &dstride)
{
    const uint32_t nrows = c->n_loc;
    const bool multi = !c->single();
    if (nrows == 0 || c->fault_after > 0) return false;
    if (!(c->glist_all && c->nblk == 0 && !c->rowsplit && (c->single() || (c->p2p && c->ll_fused)))) return false;
    hipDeviceProp_t prop;
    BICG_HIP(hipGetDeviceProperties(&prop, c->device));
    const int cus = prop.multiProcessorCount;
    // one workgroup per CU (its LDS): ranks sharing a GPU (tests) share the CUs; one CU is the helper's
    const int gmax = cus / std::max(1, c->comm->ranks_on_device) - 1;
    if (gmax < 1) return false;
    PersistPlan P;
    if (!persist_plan_host(diag, multi ? optr.data() : nullptr, multi ? ocol.data() : nullptr, multi ? oval.data() : nullptr, (unsigned)gmax, P))
        return false;
    const uint32_t nslices = P.nslices, spw = P.spw, nwg = P.nwg, grows = spw * kSliceRows;
    const uint32_t slots_used = P.win_slots, max_runs = P.max_runs, max_entries = P.max_entries;
    const std::vector<uint32_t> &pbase = P.pbase, &wptr = P.wptr;
    const std::vector<double> &pval = P.pval;
    const std::vector<unsigned short> &pslot = P.pslot, &rlen = P.rlen, &rdiag = P.rdiag;
    static_assert(sizeof(uint2) == 2 * sizeof(uint32_t), "run = two 32-bit words");
    std::vector<uint2> runs(P.runs.size() / 2 + 1);
    for (size_t i = 0; i < P.runs.size() / 2; ++i) runs[i] = make_uint2(P.runs[2 * i], P.runs[2 * i + 1]);
    PersistArgs &a = c->persist;
    a = PersistArgs{};
    a.nrows = nrows; a.nslices = nslices; a.nwg = nwg; a.spw = P.nrw; a.rpt = P.rpt;
    a.win_slots = slots_used; a.max_runs = max_runs;
    // the matrix goes to LDS when everything fits next to the window
    a.mat_entries = P.rpt == 1 ? max_entries : 0;
    if (knob_x("BICG_PERSIST_LDSMAT") && atoi(knob_x("BICG_PERSIST_LDSMAT")) == 0) a.mat_entries = 0;
    // what a workgroup may ask for on THIS device (gfx950: 160 KiB per CU; the static part of the kernels is < 6 KiB)
    const unsigned lds_max = std::min<unsigned>(kPersistMaxLds, prop.sharedMemPerBlock > 8192 ? (unsigned)prop.sharedMemPerBlock - 6144u : 0u);
    if (persist_lds_bytes(a) > lds_max) a.mat_entries = 0;
    if (persist_lds_bytes(a) > lds_max) { a = PersistArgs{}; return false; }
    auto keep = [&](void *p) { c->persist_mem.push_back(p); return p; };
    a.pval = (const double *)keep(dev_upload(pval.data(), pval.size()));
    a.pslot = (const unsigned short *)keep(dev_upload(pslot.data(), pslot.size()));
    a.pbase = (const uint32_t *)keep(dev_upload(pbase.data(), pbase.size()));
    a.rlen = (const unsigned short *)keep(dev_upload(rlen.data(), rlen.size()));
    a.rdiag = (const unsigned short *)keep(dev_upload(rdiag.data(), rdiag.size()));
    a.win_ptr = (const uint32_t *)keep(dev_upload(wptr.data(), wptr.size()));
    a.win_runs = (const uint2 *)keep(dev_upload(runs.data(), runs.size()));
    for (int i = 0; i < 4; ++i) {
        a.llv[i] = (llword *)keep(dev_alloc<llword>(2 * (size_t)nrows));
        BICG_HIP(hipMemset(a.llv[i], 0, sizeof(llword) * 2 * (size_t)nrows));
    }
    for (int i = 0; i < 2; ++i) {
        a.dtab[i] = (llword *)keep(dev_alloc<llword>((size_t)nwg * kRedSlots * 2));
        BICG_HIP(hipMemset(a.dtab[i], 0, sizeof(llword) * (size_t)nwg * kRedSlots * 2));
        a.arow[i] = (llword *)keep(dev_alloc<llword>(8));
        BICG_HIP(hipMemset(a.arow[i], 0, sizeof(llword) * 8));
        a.crow[i] = (llword *)keep(dev_alloc<llword>(6 * kPersistMaxShifts * 2));      // shifted kernel: per-shift coefficients
        BICG_HIP(hipMemset(a.crow[i], 0, sizeof(llword) * 6 * kPersistMaxShifts * 2));
    }
    a.multi = multi ? 1 : 0;
    if (multi) {
        // send-list entries by owning workgroup (the list is grouped by destination, a row may go to several ranks)
        std::vector<uint32_t> sptr(nwg + 1, 0u);
        for (uint32_t i = 0; i < c->nsend; ++i) sptr[send_idx[i] / grows + 1]++;
        for (uint32_t g = 0; g < nwg; ++g) sptr[g + 1] += sptr[g];
        std::vector<uint32_t> fill(sptr.begin(), sptr.end() - 1);
        std::vector<unsigned short> srow(c->nsend ? c->nsend : 1);
        std::vector<unsigned long long> sd0(c->nsend ? c->nsend : 1), sst(c->nsend ? c->nsend : 1);
        for (uint32_t i = 0; i < c->nsend; ++i) {
            const uint32_t g = send_idx[i] / grows, at = fill[g]++;
            srow[at] = (unsigned short)(send_idx[i] - g * grows); sd0[at] = dst0[i]; sst[at] = dstride[i];
        }
        a.snd_ptr = (const uint32_t *)keep(dev_upload(sptr.data(), sptr.size()));
        a.snd_row = (const unsigned short *)keep(dev_upload(srow.data(), srow.size()));
        a.snd_dst0 = (const unsigned long long *)keep(dev_upload(sd0.data(), sd0.size()));
        a.snd_stride = (const unsigned long long *)keep(dev_upload(sst.data(), sst.size()));
        a.ring = c->halo_ring; a.halo = c->halo;
    }
    a.v = c->v;
    a.alarm = c->alarm;
    if (getenv("BICG_DEBUG"))
        fprintf(stderr, "bicgstab_hip: rank %d: persistent plan: %u workgroups x (%u + 64) threads x %u rows (+1 helper), window %u slots (%u runs at most), "
                        "matrix %s (%u entries per workgroup), %u bytes of LDS\n", c->rank, nwg, 64 * P.nrw, P.rpt, slots_used, max_runs,
                a.mat_entries ? "in LDS" : "in memory", max_entries, persist_lds_bytes(a));
    return true;
}

// niter iterations of pipe_bicgstab in one launch (the open dot group has been closed: fetch_scal precedes every chunk)
bool persist_chunk(bicg_ctx *c, int niter)
{
    if (c->grp.active) die("internal", "persistent chunk with an open dot group");
    if (c->f1_done) die("internal", "persistent chunk after phase 1 of the next iteration has run");
    const bool plain = c->method == BICG_BICGSTAB;
    const bool pipe = c->method >= BICG_PIPE_BICGSTAB;
    const unsigned groups = plain ? 3u : 2u;                  // dot groups (tags, mailbox numbers) per iteration
    PersistArgs a = c->persist;
    a.v = c->v; a.S = c->S; a.alarm = c->alarm; a.niter = niter;
    a.seq0 = c->persist_seq;
    a.vseq0 = c->persist_vseq;
    // the pipelined kernel numbers hand-offs and groups densely and reports what it used (replacement iterations and drift
    // checks make the count data dependent): persist_account() advances the counters after the launch
    if (!pipe) c->persist_seq += groups * (unsigned)niter;
    a.it0 = c->it;
    a.krr = c->method == BICG_PIPE_BICGSTAB_RR ? c->opt.krr : 0; a.nrr = c->opt.nrr;
    a.force_first = 0;
    a.drift_every = (pipe && c->opt.rr_drift > 0.0) ? c->opt.check_every : 0;
    a.drift_tol2 = c->opt.rr_drift * c->opt.rr_drift;
    a.timeout_ticks = c->p2p ? c->p2p->timeout_ticks : 200000000ull;          // 2 s inside one GPU
    static const int xcd_map = knob_x("BICG_PERSIST_XCD") ? atoi(knob_x("BICG_PERSIST_XCD")) : 1;
    a.xcd_map = xcd_map;
    static const int first_sleep = knob_x("BICG_PERSIST_SLEEP") ? atoi(knob_x("BICG_PERSIST_SLEEP")) : 1;
    a.first_sleep = (unsigned)first_sleep;
    if (a.multi) {
        // every rank advances its exchange and group numbers by the whole chunk, converged early or not
        a.halo_seq0 = c->halo_seq;
        a.p2p = c->p2p->red_desc(c->p2p->red_seq);
        if (!pipe) { c->halo_seq += 2u * (unsigned)niter; c->p2p->red_seq += groups * (unsigned)niter; }
        a.ring = c->halo_ring;
        c->halo_unsynced = 0;
    }
    if (a.multi) {
        if (!c->waitlog) { c->waitlog = dev_alloc<unsigned>(3 * (size_t)kWaitCap); BICG_HIP(hipMemsetAsync(c->waitlog, 0, sizeof(unsigned) * 3 * kWaitCap, c->sc)); }
        a.waitlog = c->waitlog; a.waitcap = kWaitCap;
    }
    static const bool want_trace = knob_x("BICG_PERSIST_TRACE") != nullptr;
    unsigned long long *dbg = nullptr;
    if (want_trace) {
        dbg = dev_alloc<unsigned long long>(64 * 16);
        BICG_HIP(hipMemset(dbg, 0, 64 * 16 * sizeof(unsigned long long)));
        a.dbg = dbg;
    }
    hipError_t err;
    if (plain) err = launch_plain_persist(a, c->sc);
    else if (c->method == BICG_CA_BICGSTAB) err = launch_ca_persist(a, c->sc);
    else err = launch_pipe_persist(a, c->sc);
    if (err != hipSuccess) {
        // nothing ran: hand the chunk back to the multi-launch kernels (every rank sees the same failure: same kernel, same
        // plan limits; the sequence numbers reserved above are simply skipped on all of them)
        if (dbg) (void)hipFree(dbg);
        if (c->nranks > 1) die("persistent kernel", "launch failed on a multi-rank run (BICG_PERSIST=0 selects the multi-launch iteration)");
        fprintf(stderr, "bicgstab_hip: falling back to the multi-launch iteration\n");
        c->persist_on = false;
        return false;
    }
    if (want_trace && c->method != BICG_PIPE_BICGSTAB) { BICG_HIP(hipStreamSynchronize(c->sc)); BICG_HIP(hipFree(dbg)); }
    if (want_trace && c->method == BICG_PIPE_BICGSTAB) {
        // 10 ns ticks of one row workgroup (0 start, 1 z and partials published, 2 window staged, 3 product done, 4 omega here,
        // 5 w and partials published, 6 window, 7 product, 8 scalars here) and of the helper (10 / 11: group 1 / 2 published)
        std::vector<unsigned long long> h(64 * 16);
        BICG_HIP(hipStreamSynchronize(c->sc));
        BICG_HIP(hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        BICG_HIP(hipFree(dbg));
        for (int it = std::max(0, std::min(niter, 32) - 5); it < std::min(niter, 32); ++it)
            for (int who = 0; who < 2; ++who) {
                const unsigned long long *q = h.data() + (size_t)(it * 2 + who) * 16, *q0 = h.data() + (size_t)(it * 2) * 16;
                fprintf(stderr, "persist trace it %2d %s:", it, who ? "comm" : "row ");
                for (int i = 0; i <= 8; ++i) fprintf(stderr, " %d:%+.2f", i, 0.01 * (double)(long long)(q[i] - q0[0]));
                if (!who) fprintf(stderr, "  helper g1 %+.2f g2 %+.2f", 0.01 * (double)(long long)(q[10] - q0[0]), 0.01 * (double)(long long)(q[11] - q0[0]));
                else fprintf(stderr, "  helper g1: arrived %+.2f summed %+.2f applied %+.2f", 0.01 * (double)(long long)(q[12] - q0[0]),
                             0.01 * (double)(long long)(q[13] - q0[0]), 0.01 * (double)(long long)(q[14] - q0[0]));
                fprintf(stderr, "\n");
            }
    }
    return true;
}

// niter iterations of shifted_pipe_lopbicgstab (reference src/shifted_solver.c:794-866) in one launch: the seed system's
// pipelined recurrence with products of A + sigma_seed I, every other shift's p_j / x_j streamed through in phase 2. Sequence
// numbers as for the pipelined kernel (dense, reported back: persist_account).
bool persist_chunk_shifted(bicg_ctx *c, int mode, int niter, int it0, int nsig, int seed, double shift)
{
    const bool pipe = mode == SH_PIPE;          // else shifted_lopbicgstab: three groups and two products per iteration, numbered
                                                // like the plain kernel's (fixed counts, advanced here)
    if (c->grp.active) die("internal", "persistent chunk with an open dot group");
    const size_t st = c->stride;
    PersistArgs a = c->persist;
    a.v = c->v;
    a.v.x = c->x_set + (size_t)seed * st; a.v.p = c->p_set + (size_t)seed * st;      // x[seed], p[seed]
    a.S = c->S; a.alarm = c->alarm; a.niter = niter;
    a.seq0 = c->persist_seq; a.vseq0 = c->persist_vseq;
    a.it0 = it0; a.krr = 0; a.nrr = 0; a.force_first = 0; a.drift_every = 0; a.drift_tol2 = 0.0;
    a.pset = c->p_set; a.xset = c->x_set; a.set_stride = (uint32_t)st; a.nsig = nsig; a.seed = seed;
    a.shift = shift; a.has_shift = 1;
    {   // the sets stay in the Infinity Cache when they (and the matrix, if it is not in LDS) fit half of it
        const double ws = 16.0 * (double)nsig * (double)st + (a.mat_entries ? 0.0 : (double)c->matrix_bytes);
        a.set_nt = ws > 0.5 * 256.0 * 1048576.0;
        if (const char *e = knob_x("BICG_SHP_NT")) a.set_nt = atoi(e) != 0;
    }
    a.timeout_ticks = c->p2p ? c->p2p->timeout_ticks : 200000000ull;
    static const int xcd_map = knob_x("BICG_PERSIST_XCD") ? atoi(knob_x("BICG_PERSIST_XCD")) : 1;
    a.xcd_map = xcd_map;
    static const int first_sleep = knob_x("BICG_PERSIST_SLEEP") ? atoi(knob_x("BICG_PERSIST_SLEEP")) : 1;
    a.first_sleep = (unsigned)first_sleep;
    if (!pipe) c->persist_seq += 3u * (unsigned)niter;
    if (a.multi) {
        a.halo_seq0 = c->halo_seq;
        a.p2p = c->p2p->red_desc(c->p2p->red_seq);
        if (!pipe) { c->halo_seq += 2u * (unsigned)niter; c->p2p->red_seq += 3u * (unsigned)niter; }
        a.ring = c->halo_ring;
        c->halo_unsynced = 0;
    }
    const hipError_t err = pipe ? launch_shpipe_persist(a, c->sc) : launch_shlop_persist(a, c->sc);
    if (err != hipSuccess) {
        if (c->nranks > 1) die("persistent kernel", "launch failed on a multi-rank run (BICG_PERSIST=0 selects the multi-launch iteration)");
        fprintf(stderr, "bicgstab_hip: falling back to the multi-launch iteration\n");
        return false;
    }
    return true;
}

// after a pipelined persistent launch (fetch_scal has brought the scalar block back): advance the sequence counters by what
// the launch consumed. Identical on every rank -- the decisions inside the launch depend on globally reduced sums only.
void persist_account(bicg_ctx *c)
{
    const unsigned nv = (unsigned)c->hS->red[kRedUsedV], ng = (unsigned)c->hS->red[kRedUsedG];
    c->persist_seq += ng;
    c->persist_vseq += nv;
    if (!c->single()) { c->halo_seq += nv; c->p2p->red_seq += ng; }
    c->adaptive_rr += (int)c->hS->red[kRedAdaptive];
}

// vectors, reduction scratch and scalar blocks of a context whose plan (n_loc, halo, nblk) is known
static void ctx_state(bicg_ctx *c, Comm *comm, uint32_t ngroups)
{
    // ---- vectors: 12 x (rows + halo), each 256-byte aligned; order x r | rh p s y z w v t ax b
    c->stride = ((c->n_loc + c->halo + 31u) / 32u) * 32u;
    // (BICG_STRIDE_PAD = doubles added to the distance between two vectors, a multiple of 32: measurement knob for grids whose
    // vectors would otherwise lie a power of two bytes apart -- 512^3: exactly 1 GiB)
    if (const char *sv = knob_x("BICG_STRIDE_PAD")) c->stride += ((uint32_t)std::max(0, atoi(sv)) / 32u) * 32u;
    c->slab = dev_alloc<double>(12 * (size_t)c->stride);
    BICG_HIP(hipMemset(c->slab, 0, sizeof(double) * 12 * (size_t)c->stride));
    double *base = c->slab;
    double **slots[12] = {&c->v.x, &c->v.r, &c->v.rh, &c->v.p, &c->v.s, &c->v.y, &c->v.z, &c->v.w, &c->v.v, &c->v.t, &c->v.ax, &c->v.b};
    for (int i = 0; i < 12; ++i) *slots[i] = base + (size_t)i * c->stride;
    c->v.n = c->n_loc;

    c->nslots = std::max<unsigned>(ngroups + c->nblk, kMaxGrid) + 64;
    c->partial = dev_alloc<double>((size_t)c->nslots * kPartialStride);
    c->shard_tot = dev_alloc<double>((size_t)kShards * kPartialStride);
    c->counter = dev_alloc<unsigned>((kShards + 1) * kCounterStride);
    BICG_HIP(hipMemset(c->counter, 0, sizeof(unsigned) * (kShards + 1) * kCounterStride));
    c->tail_tab = dev_alloc<llword>((size_t)c->nslots * kTailStride);
    BICG_HIP(hipMemset(c->tail_tab, 0, sizeof(llword) * (size_t)c->nslots * kTailStride));
    c->tail_shard = dev_alloc<llword>((size_t)kShards * kRedSlots * 2);
    BICG_HIP(hipMemset(c->tail_shard, 0, sizeof(llword) * kShards * kRedSlots * 2));
    if (const char *sv = knob_x("BICG_TAIL_FINISH")) c->tail_finish = atoi(sv) != 0;
    c->Sbuf = dev_alloc<Scal>(2);
    BICG_HIP(hipMemset(c->Sbuf, 0, 2 * sizeof(Scal)));
    c->S = c->Sbuf;
    for (int i = 0; i < 2; ++i) {
        c->wpart[i] = dev_alloc<double>((size_t)c->nslots * (kBlock / 64) * kPartialStride);
        BICG_HIP(hipMemset(c->wpart[i], 0, sizeof(double) * (size_t)c->nslots * (kBlock / 64) * kPartialStride));
    }
    c->shard_ll = dev_alloc<llword>((size_t)2 * kShardLL * kRedSlots * 2);
    BICG_HIP(hipMemset(c->shard_ll, 0, sizeof(llword) * 2 * kShardLL * kRedSlots * 2));
    c->alarm = dev_alloc<int>(1);
    BICG_HIP(hipMemset(c->alarm, 0, sizeof(int)));
    BICG_HIP(hipHostMalloc((void **)&c->h_alarm, sizeof(int), hipHostMallocDefault));
    *c->h_alarm = 0;
    if (comm->ranks_on_device > 1) {
        // one-GPU box standing in for a node: 1024 = 256 CUs x 4 resident workgroups of the largest kernels
        c->wg_cap = 1024u / (unsigned)(comm->ranks_on_device + 1);
        set_vec_grid_cap(c->wg_cap);
    }
}

static void ctx_streams(bicg_ctx *c, int P)
{
    BICG_HIP(hipStreamCreateWithFlags(&c->sc, hipStreamNonBlocking));
    if (P > 1 || c->force_comm) BICG_HIP(hipStreamCreateWithFlags(&c->sm, hipStreamNonBlocking));
    for (int i = 0; i < kEvRing; ++i) {
        BICG_HIP(hipEventCreateWithFlags(&c->ev_pack[i], hipEventDisableTiming));
        BICG_HIP(hipEventCreateWithFlags(&c->ev_halo[i], hipEventDisableTiming));
        BICG_HIP(hipEventCreateWithFlags(&c->ev_dots[i], hipEventDisableTiming));
        BICG_HIP(hipEventCreateWithFlags(&c->ev_red[i], hipEventDisableTiming));
    }
    BICG_HIP(hipDeviceSynchronize());       // uploads and memsets above used the null stream
}

// =====================================================================================  C ABI
extern "C" {

int bicg_has_experiments(void) { return kExperiments ? 1 : 0; }
int bicg_switch_unknown(const char *set, char *out, int cap) { return set ? knob_unknown(set, out, cap > 0 ? (size_t)cap : 0) : 0; }
// once per process, from bicg_create: a token none of the lists knows is reported, not obeyed
static void warn_unknown_switches()
{
    static bool done = false;
    if (done) return;
    done = true;
    for (const char *set : {"BICG_PLAN", "BICG_PERSIST", "BICG_TEST"}) {
        char first[64];
        const int n = knob_unknown(set, first, sizeof first);
        if (n) fprintf(stderr, "bicgstab_hip: %s has %d token%s this library does not know (first: \"%s\"); see INTEGRATION.md section 6\n", set, n, n == 1 ? "" : "s", first);
    }
}
int bicg_switch_value(const char *set, const char *name, char *out, int cap)
{
    const char *v = set && name ? knob_tok(set, name) : nullptr;
    if (!v) return -1;
    if (out && cap > 0) { strncpy(out, v, (size_t)cap - 1); out[cap - 1] = 0; }
    return (int)strlen(v);
}
const char *bicg_version(void) { return "bicgstab_hip 0.1 (gfx950)"; }

void bicg_default_options(bicg_options *o)
{
    memset(o, 0, sizeof *o);
    o->tol = 1.0e-15;      // reference EPS       (src/solver.c:3)
    o->max_iter = 1000;    // reference MAX_ITER  (src/solver.c:4)
    o->out_iter = 100;     // reference OUT_ITER  (src/solver.c:9)
    o->check_every = 16;
}

// the code objects this context launches from, loaded now (preload_kernels, bicg_kernels.hip)
static void preload_for(bicg_ctx *c)
{
    if (knob_x("BICG_PRELOAD") && atoi(knob_x("BICG_PRELOAD")) == 0) return;
    SellDev d = {c->s_val, c->s_col, c->s_base, c->s_len, c->s_col16, c->s_base16, c->sell_jag ? 1 : 0, c->win_ptr, c->win_runs, c->win_slots, c->sell_perm};
    d.vbase = c->s_vbase;
    preload_kernels(d, c->sell_entries > 0);
    if (c->persist_on) preload_persist_kernels();
    if (c->st.on) preload_stencil_kernels();
    if (c->lane_info && c->jagw_fast) preload_jagw_kernels();
}

bicg_ctx *bicg_create(const CSR_Matrix *diag, const CSR_Matrix *offd, const INFO_Matrix *info)
{
    Comm *comm = comm_get();
    BICG_HIP(hipSetDevice(comm->device));
    if (comm->rank == 0) warn_unknown_switches();
    if (info->rows != info->cols) { fprintf(stderr, "ERROR: bicg_create: matrix is not square\n"); return nullptr; }

    bicg_ctx *c = new bicg_ctx;
    c->comm = comm; c->device = comm->device; c->nranks = comm->nranks; c->rank = comm->rank;
    g_live.push_back(c);
    // a rank without rows: one phantom row (see bicg_ctx::phantom)
    static double ph_val[1] = {1.0};
    static unsigned ph_col[1] = {0u}, ph_ptr1[2] = {0u, 1u}, ph_ptr0[2] = {0u, 0u};
    CSR_Matrix ph_d, ph_o;
    if (diag->rows == 0 && info->rows > 0 && comm->nranks > 1) {
        c->phantom = true;
        ph_d.val = ph_val; ph_d.col = ph_col; ph_d.ptr = ph_ptr1; ph_d.nz = 1; ph_d.rows = 1; ph_d.cols = 1;
        ph_o.val = ph_val; ph_o.col = ph_col; ph_o.ptr = ph_ptr0; ph_o.nz = 0; ph_o.rows = 1; ph_o.cols = info->cols;
        diag = &ph_d; offd = &ph_o;
    }
    c->n_loc = diag->rows; c->n_glob = info->rows;
    c->nnz_d = diag->rows ? diag->ptr[diag->rows] : 0u;
    const int P = c->nranks;

    bool use_sell = !(knob_x("BICG_NO_SELL") && atoi(knob_x("BICG_NO_SELL")));
    if (const char *sv = knob_x("BICG_SELL_NT")) c->sell_nt_env = atoi(sv);
    if (const char *sv = knob_x("BICG_SELL_ALT")) c->sell_alt = atoi(sv);
    if (const char *sv = knob_x("BICG_SELL_XCD")) c->sell_xcd = atoi(sv);
    if (const char *sv = test_tok("force-comm")) c->force_comm = atoi(sv) != 0;
    if (const char *sv = getenv("BICG_GRAPH")) c->graph_mode = atoi(sv);
    uint64_t nnz_diag_all = c->nnz_d;      // diag non-zeros of all ranks
    {   // Every rank learns every rank's (non-zeros, rows). The enqueue mode changes the ORDER of RCCL calls,
        // so all ranks must take the same decision: it is based on the average number of local non-zeros.
        // (a rank without rows carries a phantom row and counts as a rank like any other; only an EMPTY MATRIX is refused)
        uint64_t total = c->nnz_d;
        bool empty = c->n_loc == 0;
        if (P > 1) {
            std::vector<int> cnt(P, 2 * (int)sizeof(uint32_t)), off(P);
            std::vector<uint32_t> mine(2 * (size_t)P), all(2 * (size_t)P, 0u);
            for (int p = 0; p < P; ++p) { off[p] = 2 * p * (int)sizeof(uint32_t); mine[2 * p] = c->nnz_d; mine[2 * p + 1] = c->n_loc; }
            comm->alltoallv_host(mine.data(), cnt.data(), off.data(), all.data(), cnt.data(), off.data());
            all[2 * c->rank] = c->nnz_d; all[2 * c->rank + 1] = c->n_loc;
            total = 0;
            for (int p = 0; p < P; ++p) { total += all[2 * p]; empty = empty || all[2 * p + 1] == 0; }
        }
        if (empty) {
            if (c->rank == 0) fprintf(stderr, "ERROR: bicg_create: empty matrix (%u rows over %d ranks)\n", info->rows, P);
            bicg_destroy(c);          // nothing is allocated yet; takes the context out of the registry of live ones
            return nullptr;
        }
        nnz_diag_all = total;
        c->overlap = total / (uint64_t)P >= 6000000u;
        // two launches per pipelined iteration (phases in the SpMV epilogues): latency on small ranks (200 k rows 26.2
        // vs 34.1 us), the traffic of v and t on large ones (1.6 M rows 159 vs 163 us, banded b = 8 158 vs 169, the
        // 16.8 M-row Laplacian share 1.14 vs 1.25 ms) -- except with x windows, whose epilogue kernels at 4 waves per
        // SIMD lose on large blocks (FEM-like 189 vs 175 us). Like the enqueue mode this changes the sequence of
        // exchanges, so it is decided from facts all ranks share (see fuse_plan_ok), never from the local block alone.
        c->fuse_small = total / (uint64_t)P < 6000000u;
    }
    if (const char *sv = getenv("BICG_OVERLAP")) c->overlap = atoi(sv) != 0;
    if (const char *sv = knob_x("BICG_SELL_GPW")) c->sell_gpw = atoi(sv);
    if (const char *sv = knob_x("BICG_SELL_GPW_DOTS")) c->sell_gpw_dots = atoi(sv);

    // ---- halo plan (multi rank): which of x's remote entries this rank needs, who needs ours
    std::vector<uint32_t> ocol, optr(c->n_loc + 1, 0u);
    std::vector<double> oval;
    std::vector<uint32_t> send_idx;
    c->scnt.assign(P, 0); c->sdsp.assign(P, 0); c->rcnt.assign(P, 0); c->rdsp.assign(P, 0);
    if (P > 1) {
        if (offd->rows != c->n_loc) die("bicg_create", "offd block row count differs from diag block");
        c->nnz_o = offd->ptr[offd->rows];
        std::vector<uint32_t> halo_cols(c->nnz_o ? c->nnz_o : 1);
        ocol.resize(c->nnz_o ? c->nnz_o : 1);
        c->halo = (uint32_t)bicg_halo_plan(offd, info, P, c->n_loc, halo_cols.data(), c->rcnt.data(), ocol.data());
        optr.assign(offd->ptr, offd->ptr + c->n_loc + 1);
        oval.assign(offd->val, offd->val + c->nnz_o);
        for (int p = 1; p < P; ++p) c->rdsp[p] = c->rdsp[p - 1] + c->rcnt[p - 1];
        // tell every owner which of its rows we need; learn which of ours the others need
        auto tramp = [](const void *sbuf, const int *sc, const int *sd, void *rbuf, const int *rc, const int *rd, void *user) {
            static_cast<Comm *>(user)->alltoallv_host(sbuf, sc, sd, rbuf, rc, rd);
        };
        const int total = bicg_halo_send_counts(P, c->rcnt.data(), tramp, comm, c->scnt.data());
        send_idx.resize(total > 0 ? total : 1);
        const int got = bicg_halo_send_lists(c->rank, P, info, c->n_loc, halo_cols.data(), c->rcnt.data(), c->scnt.data(),
                                             tramp, comm, send_idx.data());
        if (got < 0) die("bicg_create", "halo request outside the owner's rows");
        c->nsend = (uint32_t)got;
        for (int p = 1; p < P; ++p) c->sdsp[p] = c->sdsp[p - 1] + c->scnt[p - 1];
    }

    // (BICG_PLAN_TRACE=1: seconds per part of the plan on stderr, rank 0)
    const bool plan_trace = getenv("BICG_PLAN_TRACE") && atoi(getenv("BICG_PLAN_TRACE")) != 0 && comm->rank == 0;
    double plan_t = now_sec();
    auto plan_mark = [&](const char *what) {
        if (!plan_trace) return;
        const double t = now_sec();
        fprintf(stderr, "bicgstab_hip: plan  %-34s %8.4f s\n", what, t - plan_t);
        plan_t = t;
    };
    plan_mark("state, halo plan");
    // ---- SpMV plan. Rows are cut into groups of 256 (4 slices of 64 rows = one workgroup, lane = row).
    // Two layouts of a slice: PADDED to its longest row (banded matrices: nothing to pad, 8-byte loads of four
    // 16-bit column offsets) or JAGGED (ragged rows: step k stores the rows longer than k only; exactly the CSR's
    // bytes, lane = row kept). Jagged is chosen for the whole block when padding would add > 2 % entries. Groups
    // with a very long row go to the CSR row-block kernel (strided workgroup reduction of one row). Either kind
    // is "boundary" when one of its rows has offd entries (it then runs after the halo has landed).
    const uint32_t nrows = c->n_loc;
    const uint32_t nslices = (nrows + kSliceRows - 1) / kSliceRows, ngroups = (nrows + kGroupRows - 1) / kGroupRows;
    // Long rows: lane = row needs 256 rows per workgroup, so a block of few, long rows (banded, half-bandwidth 512:
    // 23 k rows of 1025 entries = 92 workgroups for 256 CUs) starves the GPU. Such a block goes to the rows-over-lanes
    // kernel (k_spmv_rows) as a whole: row blocks of <= 8192 non-zeros, a row spread over 8..64 lanes. The row sums
    // are then associated differently from mult() (tolerance 1e-13 x sum |a_ij x_j| instead of bit-exact).
    // Decided from the GLOBAL shape (mean row length, rows per rank) so that all ranks agree.
    {
        const uint64_t mean_len = info->rows ? nnz_diag_all / info->rows : 0;     // (INFO_Matrix.nz is not always filled in)
        const uint64_t groups_per_rank = ((uint64_t)info->rows / (uint64_t)P + kGroupRows - 1) / kGroupRows;
        c->rowsplit = use_sell && (mean_len >= 256 || (mean_len >= 128 && groups_per_rank < 512));
        if (const char *sv = knob_x("BICG_ROWSPLIT")) c->rowsplit = atoi(sv) != 0;
        if (c->rowsplit) use_sell = false;
    }
    std::vector<uint32_t> slice_len(nslices, 0u), slice_base(nslices, 0u);
    for (uint32_t r = 0; r < nrows; ++r)
        slice_len[r / kSliceRows] = std::max(slice_len[r / kSliceRows], diag->ptr[r + 1] - diag->ptr[r]);
    std::vector<uint32_t> gl_int, gl_bnd;
    std::vector<uint4> bint, bbnd;
    std::vector<char> group_is_sell(ngroups, 0);
    const uint32_t jag_max_row = std::max<uint64_t>(64, nrows ? 4 * (uint64_t)c->nnz_d / nrows : 0);   // 4 x the average row
    bool jag = false;
    {
        uint64_t padded_rows = 0;
        for (uint32_t sl = 0; sl < nslices; ++sl)
            padded_rows += (uint64_t)slice_len[sl] * std::min<uint32_t>(kSliceRows, nrows - sl * kSliceRows);
        jag = padded_rows > (uint64_t)c->nnz_d + c->nnz_d / 50;
        if (const char *sv = plan_tok("layout")) jag = !strcmp(sv, "jag") ? true : !strcmp(sv, "pad") ? false : jag;
    }
    // x windows in LDS (SellDev::win_*): wanted for ragged rows, where the x gather of one step touches many cache
    // lines (FEM-like: 63 -> 58 us per SpMV). With equal rows the gathers are perfectly coalesced and the window
    // only adds staging loads and two barriers per group (Transport-shaped +2 %, 256^3 Laplacian +9 % although its
    // columns shrink from 32 to 16 bits), so there it is taken on request only: BICG_PLAN="window=1" asks for it
    // whenever it fits, 0 never. It needs the jagged layout.
    const bool jag_auto = jag;
    int win_env = -1;
    if (const char *sv = plan_tok("window")) win_env = atoi(sv);
    bool want_win = use_sell && win_env != 0 && (win_env == 1 || jag_auto);
    if (want_win) jag = true;
    auto group_fits = [&](uint32_t g, uint64_t *stored_out) {
        const uint32_t r0 = g * kGroupRows, r1 = std::min(nrows, r0 + kGroupRows);
        const uint64_t nnz_g = diag->ptr[r1] - diag->ptr[r0];
        if (jag) {
            // a lane walks its row alone: an outlier row would keep its wavefront busy long after the launch's other
            // rows are done, so it goes to the CSR kernel, which spreads one row over a workgroup
            *stored_out = nnz_g;
            for (uint32_t sl = r0 / kSliceRows; sl * kSliceRows < r1; ++sl)
                if (slice_len[sl] > jag_max_row) return false;
            return true;
        }
        // storage always covers 64 lanes per slice; the criterion only counts lanes that hold a row, so
        // that the last, partly filled group of a block does not fall to the CSR kernel (an extra
        // launch per SpMV for a few dozen rows)
        uint64_t padded = 0, padded_rows = 0;
        for (uint32_t sl = r0 / kSliceRows; sl * kSliceRows < r1; ++sl) {
            padded += (uint64_t)slice_len[sl] * kSliceRows;
            padded_rows += (uint64_t)slice_len[sl] * std::min<uint32_t>(kSliceRows, r1 - sl * kSliceRows);
        }
        *stored_out = padded;
        return padded_rows <= nnz_g + nnz_g / 4 + 2 * kSliceRows;
    };
    // (Round 1, before the jagged layout: a ragged matrix left only a few groups under the padding limit; two
    // kernels per SpMV were then slower than the CSR kernel alone -- synth.fem_like 70 vs 63 us -- and sorting rows
    // by length inside the groups, SELL-C-sigma, removes the padding but also the coalesced x gather: 66.9 us.)
    bool sell_worthwhile = use_sell;
    uint64_t sell_entries = 0;
    std::vector<uint32_t> win_ptr;
    std::vector<uint2> win_runs;
    uint32_t win_slots = 0;
    bool win_list_mode = false;
  select_groups:
    sell_entries = 0; c->sell_nnz = 0; c->sell_rows = 0;
    gl_int.clear(); gl_bnd.clear();
    if (use_sell) {
        uint64_t rows_fit = 0, dummy;
        for (uint32_t g = 0; g < ngroups; ++g)
            if (group_fits(g, &dummy)) rows_fit += std::min(nrows, (g + 1) * (uint32_t)kGroupRows) - g * kGroupRows;
        sell_worthwhile = 2 * rows_fit >= nrows;
    }
    for (uint32_t g = 0; g < ngroups; ++g) {
        const uint32_t r0 = g * kGroupRows, r1 = std::min(nrows, r0 + kGroupRows);
        const uint64_t nnz_g = diag->ptr[r1] - diag->ptr[r0];
        uint64_t stored = 0;
        const bool sell = sell_worthwhile && group_fits(g, &stored) && sell_entries + stored < 0xFFFFFF00ull;
        group_is_sell[g] = sell;
        if (!sell) continue;
        for (uint32_t sl = r0 / kSliceRows; sl * kSliceRows < r1; ++sl) {
            slice_base[sl] = (uint32_t)sell_entries;
            if (jag) sell_entries += diag->ptr[std::min(nrows, (sl + 1) * (uint32_t)kSliceRows)] - diag->ptr[sl * kSliceRows];
            else sell_entries += (uint64_t)slice_len[sl] * kSliceRows;
        }
        c->sell_nnz += nnz_g; c->sell_rows += r1 - r0;
        const bool touches_halo = P > 1 && optr[r1] > optr[r0];
        (touches_halo ? gl_bnd : gl_int).push_back(g);
    }
    if (want_win) {
        // per group: the columns its rows touch, merged into runs of consecutive columns (bicg_plan.cpp)
        constexpr uint32_t kWinGap = 8;
        bool ok = sell_entries > 0;
        long nruns = ok ? bicg_window_plan(diag->ptr, diag->col, nrows, kGroupRows, group_is_sell.data(), kWinMaxSlots, kWinGap,
                                           nullptr, nullptr, nullptr) : -1;
        if (nruns >= 0) {
            win_ptr.assign(ngroups + 1, 0u);
            win_runs.assign((size_t)nruns + 1, make_uint2(0u, 0u));
            static_assert(sizeof(uint2) == 2 * sizeof(unsigned int), "run = two 32-bit words");
            bicg_window_plan(diag->ptr, diag->col, nrows, kGroupRows, group_is_sell.data(), kWinMaxSlots, kWinGap, win_ptr.data(),
                             reinterpret_cast<unsigned int *>(win_runs.data()), &win_slots);
        } else {
            ok = false;
        }
        // The window pays through k_spmv_jagw only (three dependent trips per group, bicg_jagw.hip): at most kJagwMaxRuns runs per
        // group and kJagwMaxSlots slots. A numbering whose groups touch MANY short runs (reverse Cuthill-McKee of a tetrahedral
        // mesh: up to 170 runs, 2 657 slots) would go through k_spmv_sell's window loop, which stages run after run: 169 us per
        // product on the 1.6 M-row mesh matrix against 56.5 us for the same jagged slices with 16-bit offsets gathered through
        // the caches (profiles/r06/mesh_probe_baseline.txt, mesh_probe_plans.txt). Unless BICG_PLAN="window=1" insists, such a
        // block keeps its jagged slices and drops the window.
        if (ok && win_env != 1) {
            uint32_t most_runs = 0;
            for (uint32_t g = 0; g < ngroups; ++g) most_runs = std::max(most_runs, win_ptr[g + 1] - win_ptr[g]);
            if (most_runs > kJagwMaxRuns || win_slots > kJagwMaxSlots) ok = false;
            // ... unless the group's DISTINCT columns fit the window one by one (no gaps merged): the list-driven window of
            // k_spmv_jagw<.., LIST> (SellDev::win_list). One rank only -- launches with offd entries or the exchange inside go
            // through k_spmv_sell's loop, which would have to stage hundreds of runs -- and not for blocks whose pipelined
            // phases ride in the products' epilogues (the same loop). BICG_PLAN="window-list=0" keeps the gathers.
            if (!ok && P == 1 && !c->fuse_small && sell_entries > 0 && !plan_off("window-list")) {
                long nr = bicg_window_plan(diag->ptr, diag->col, nrows, kGroupRows, group_is_sell.data(), kJagwMaxSlots, 0u, nullptr, nullptr, nullptr);
                bool near = nr >= 0;
                for (uint32_t g = 0; near && g < ngroups; ++g) {      // 16-bit list entries: distance from the group's first row
                    if (!group_is_sell[g]) continue;
                    const uint32_t r0 = g * kGroupRows, r1 = std::min(nrows, r0 + kGroupRows);
                    for (uint32_t j = diag->ptr[r0]; j < diag->ptr[r1]; ++j) {
                        const long d = (long)diag->col[j] - (long)r0;
                        if (d < -32768 || d > 32767) { near = false; break; }
                    }
                }
                if (near) {
                    win_ptr.assign(ngroups + 1, 0u);
                    win_runs.assign((size_t)nr + 1, make_uint2(0u, 0u));
                    bicg_window_plan(diag->ptr, diag->col, nrows, kGroupRows, group_is_sell.data(), kJagwMaxSlots, 0u, win_ptr.data(),
                                     reinterpret_cast<unsigned int *>(win_runs.data()), &win_slots);
                    win_list_mode = true;
                    ok = true;
                }
            }
        }
        if (!ok) {                          // some group's window does not fit: no windows for this block
            want_win = false; win_slots = 0; win_runs.clear(); win_ptr.clear();
            if (!jag_auto) { jag = false; std::fill(group_is_sell.begin(), group_is_sell.end(), 0); goto select_groups; }
        }
    }
    const bool win = want_win && win_slots > 0;
    auto slot_of = [&](uint32_t g, uint32_t col) -> uint32_t {
        return bicg_window_slot(reinterpret_cast<const unsigned int *>(win_runs.data()), win_ptr[g], win_ptr[g + 1], col);
    };
    // With windows: deal the rows of every group to the lanes by decreasing length (SellDev::perm). The group's
    // entries stay where they are as a whole; the slices inside it change length.
    std::vector<unsigned char> perm;
    if (win && !(knob_x("BICG_SELL_SORT") && atoi(knob_x("BICG_SELL_SORT")) == 0)) {
        perm.assign((size_t)ngroups * kGroupRows, 0);
        std::vector<uint32_t> slice_sum(nslices, 0u);             // entries of a slice after the rows were dealt out
        parallel_ranges(ngroups, 64, [&](size_t ga, size_t gb, int) {
            for (uint32_t g = (uint32_t)ga; g < (uint32_t)gb; ++g) {
                unsigned char *pg = perm.data() + (size_t)g * kGroupRows;
                for (uint32_t t = 0; t < kGroupRows; ++t) pg[t] = (unsigned char)t;
                if (!group_is_sell[g]) continue;
                const uint32_t r0 = g * kGroupRows;
                auto len_of = [&](unsigned t) -> uint32_t { return r0 + t < nrows ? diag->ptr[r0 + t + 1] - diag->ptr[r0 + t] : 0u; };
                std::stable_sort(pg, pg + kGroupRows, [&](unsigned char x, unsigned char y) { return len_of(x) > len_of(y); });
                for (uint32_t w = 0; w < kGroupRows / kSliceRows; ++w) {
                    const uint32_t sl = g * (kGroupRows / kSliceRows) + w;
                    if (sl >= nslices) break;
                    uint32_t longest = 0; uint64_t sum = 0;
                    for (uint32_t l = 0; l < kSliceRows; ++l) { const uint32_t n = len_of(pg[w * kSliceRows + l]); longest = std::max(longest, n); sum += n; }
                    slice_len[sl] = longest; slice_sum[sl] = (uint32_t)sum;
                }
            }
        });
        uint64_t at = 0;
        for (uint32_t g = 0; g < ngroups; ++g) {
            if (!group_is_sell[g]) continue;
            for (uint32_t sl = g * (kGroupRows / kSliceRows); sl < std::min(nslices, (g + 1) * (kGroupRows / kSliceRows)); ++sl) { slice_base[sl] = (uint32_t)at; at += slice_sum[sl]; }
        }
        if (at != sell_entries) die("bicg_create", "internal: sorted slices do not add up");
    }
    auto row_of = [&](uint32_t sl, uint32_t lane) -> uint32_t {      // the row lane `lane` of slice `sl` works on
        if (perm.empty()) return sl * kSliceRows + lane;
        const uint32_t g = sl / (kGroupRows / kSliceRows), w = sl % (kGroupRows / kSliceRows);
        return g * kGroupRows + perm[(size_t)g * kGroupRows + w * kSliceRows + lane];
    };
    plan_mark("groups, windows, row order");
    c->sell_entries = sell_entries;
    c->sell_jag = jag && sell_entries > 0;
    // (allocated without a fill: the threads that write a slice also zero its padding -- 330 MB of zeros from one thread were a
    // third of this part)
    std::unique_ptr<double[]> sval_mem(new double[sell_entries ? sell_entries : 1]);
    double *const sval = sval_mem.get();
    std::unique_ptr<uint32_t[]> scol_mem;                          // filled once it is known whether the 32-bit columns are uploaded
    // 16-bit column offsets when every sliced-ELL entry is within +-32767 of its row
    bool c16 = sell_entries > 0 && (win || !plan_off("col16"));
    std::vector<uint32_t> slice_base16(nslices, 0u);
    uint64_t n16 = 0;
    if (jag) n16 = sell_entries;
    else
        for (uint32_t sl = 0; sl < nslices; ++sl) {
            slice_base16[sl] = (uint32_t)n16;
            if (group_is_sell[sl / (kGroupRows / kSliceRows)]) n16 += (uint64_t)((slice_len[sl] + 3) / 4) * 4 * kSliceRows;
        }
    if (n16 >= 0xFFFFFF00ull) c16 = false;
    std::vector<int> offsets_seen;          // distinct column offsets (col - row), while they stay few: the fused-window clusters
    bool offsets_few = true;
    if (c16 && !win) {
        // row ranges on several threads, a map of the offsets seen per thread; merged below (ascending: the order does not matter,
        // the clusters are formed from the sorted list)
        std::vector<std::vector<unsigned char>> marks((size_t)plan_threads());
        std::vector<char> bad((size_t)plan_threads(), 0);
        const int np = parallel_ranges(nrows, 4096, [&](size_t ra, size_t rb, int part) {
            std::vector<unsigned char> &mark = marks[(size_t)part];
            mark.assign(65536, 0);
            for (uint32_t r = (uint32_t)ra; r < (uint32_t)rb && !bad[(size_t)part]; ++r) {
                if (!group_is_sell[r / kGroupRows]) continue;
                for (uint32_t j = diag->ptr[r]; j < diag->ptr[r + 1]; ++j) {
                    const int64_t dlt = (int64_t)diag->col[j] - (int64_t)r;
                    if (dlt < -32767 || dlt > 32767) { bad[(size_t)part] = 1; break; }
                    mark[dlt + 32768] = 1;
                }
            }
        });
        for (int p = 0; p < np; ++p) if (bad[(size_t)p]) c16 = false;
        for (int d = 0; c16 && d < 65536; ++d) {
            bool any = false;
            for (int p = 0; p < np && !any; ++p) any = marks[(size_t)p][(size_t)d] != 0;
            if (!any) continue;
            if (offsets_seen.size() >= 4096) { offsets_few = false; break; }
            offsets_seen.push_back(d - 32768);
        }
    }
    // Fused-window clusters (struct FusedWindow): the offsets fall into <= 4 clusters (gaps of more than 512 columns separate
    // them) and a group's window -- 256 + span columns per cluster -- fits 2048 LDS slots. Padded slices with 16-bit offsets,
    // every row on the sliced-ELL path. (The fused product itself is a one-rank form; the windowed SpMM uses the clusters on every rank.)
    if (c16 && !jag && !win && offsets_few && sell_entries > 0) {
        offsets_seen.push_back(0);
        std::sort(offsets_seen.begin(), offsets_seen.end());
        FusedWindow f{};
        int ncl = 0, slots = 0;
        bool ok = true;
        for (size_t i = 0; i < offsets_seen.size() && ok;) {
            size_t k = i;
            while (k + 1 < offsets_seen.size() && offsets_seen[k + 1] - offsets_seen[k] <= 512) ++k;
            if (ncl == kFwMaxClusters) { ok = false; break; }
            f.lo[ncl] = offsets_seen[i]; f.hi[ncl] = offsets_seen[k];
            f.bias[ncl] = slots - f.lo[ncl];
            slots += kGroupRows + f.hi[ncl] - f.lo[ncl];
            ++ncl;
            i = k + 1;
        }
        if (ok && slots <= 2048) { f.ncl = ncl; f.slots = (unsigned)slots; c->fw = f; }
    }
    plan_mark("column offsets, clusters");
    const size_t n16_alloc = c16 ? (size_t)n16 : 1;
    std::unique_ptr<short[]> scol16_mem(new short[n16_alloc]);
    short *const scol16 = scol16_mem.get();
    if (!c16) { scol16[0] = 0; scol_mem.reset(new uint32_t[sell_entries ? sell_entries : 1]); }
    uint32_t *const scol = scol_mem.get();                        // null with 16-bit offsets: the 32-bit columns are not uploaded
    if (sell_entries == 0) { sval[0] = 0.0; if (scol) scol[0] = 0u; }
    // Slices on several threads: a slice's entries (and its padding, zeros) are its own range of the arrays.
    parallel_ranges(nslices, 256, [&](size_t sa, size_t sb, int) {
        for (uint32_t sl = (uint32_t)sa; sl < (uint32_t)sb; ++sl) {
            const uint32_t g = sl / (kGroupRows / kSliceRows);
            if (!group_is_sell[g]) continue;
            if (jag) {
                size_t e = slice_base[sl];
                for (uint32_t k = 0; k < slice_len[sl]; ++k)
                    for (uint32_t lane = 0; lane < kSliceRows; ++lane) {
                        const uint32_t r = row_of(sl, lane);
                        if (r >= nrows || diag->ptr[r + 1] - diag->ptr[r] <= k) continue;
                        const uint32_t j = diag->ptr[r] + k;
                        sval[e] = diag->val[j];
                        if (scol) scol[e] = diag->col[j];
                        if (win) scol16[e] = (short)(unsigned short)slot_of(g, diag->col[j]);
                        else if (c16) scol16[e] = (short)((int64_t)diag->col[j] - (int64_t)r);
                        ++e;
                    }
                continue;
            }
            const size_t b0 = slice_base[sl], n = (size_t)slice_len[sl] * kSliceRows;
            std::fill(sval + b0, sval + b0 + n, 0.0);
            if (scol) std::fill(scol + b0, scol + b0 + n, 0u);
            if (c16) std::fill(scol16 + slice_base16[sl], scol16 + slice_base16[sl] + (size_t)((slice_len[sl] + 3) / 4) * 4 * kSliceRows, (short)0);
            for (uint32_t lane = 0; lane < kSliceRows; ++lane) {
                const uint32_t r = sl * kSliceRows + lane;
                if (r >= nrows) break;
                for (uint32_t j = diag->ptr[r], k = 0; j < diag->ptr[r + 1]; ++j, ++k) {
                    const size_t e = b0 + (size_t)k * kSliceRows + lane;
                    sval[e] = diag->val[j];
                    if (scol) scol[e] = diag->col[j];
                    if (c16) scol16[(size_t)slice_base16[sl] + ((size_t)(k / 4) * kSliceRows + lane) * 4 + (k % 4)] =
                                 (short)((int64_t)diag->col[j] - (int64_t)r);
                }
            }
        }
    });

    plan_mark("sliced-ELL arrays");
    // Uniform slices (SellDev::ubase): all 64 rows present, equally long, entry k at the same distance from its row in
    // every row. Lists are shared between slices (a banded matrix has ONE for its whole interior) and padded with zeros.
    std::vector<uint32_t> ubase, vbase, mbase;
    std::vector<int> uoff;
    std::vector<double> uval;
    std::vector<unsigned short> rmask;
    uint64_t uniform_entries = 0, constant_entries = 0, masked_rows = 0;
    const bool want_constant = !plan_off("constant");
    const bool want_masked = !plan_off("masked");
    if (!jag && sell_entries > 0 && !plan_off("uniform")) {
        ubase.assign(nslices, 0xFFFFFFFFu);
        std::map<std::vector<int>, uint32_t> lists, vlists;
        std::vector<int> cur, vkey;
        // which slices are uniform (1) / uniform and constant (2): 64 rows x length comparisons per slice, on several threads; the
        // lists themselves are numbered by the pass below, in slice order
        std::vector<char> cls(nslices, 0);
        parallel_ranges(nslices, 256, [&](size_t sa, size_t sb, int) {
            for (uint32_t sl = (uint32_t)sa; sl < (uint32_t)sb; ++sl) {
                if (!group_is_sell[sl / (kGroupRows / kSliceRows)] || (sl + 1) * kSliceRows > nrows || slice_len[sl] == 0) continue;
                const uint32_t r0 = sl * kSliceRows, len = slice_len[sl], p0 = diag->ptr[r0];
                bool uni = true;
                for (uint32_t l = 0; l < kSliceRows && uni; ++l) uni = diag->ptr[r0 + l + 1] - diag->ptr[r0 + l] == len;
                for (uint32_t l = 1; l < kSliceRows && uni; ++l)
                    for (uint32_t k = 0; k < len; ++k)
                        if ((int64_t)diag->col[diag->ptr[r0 + l] + k] - (int64_t)(r0 + l) != (int64_t)diag->col[p0 + k] - (int64_t)r0) { uni = false; break; }
                if (!uni) continue;
                bool con = want_constant;
                for (uint32_t l = 1; l < kSliceRows && con; ++l) con = memcmp(diag->val + diag->ptr[r0 + l], diag->val + p0, sizeof(double) * len) == 0;
                cls[sl] = con ? 2 : 1;
            }
        });
        for (uint32_t sl = 0; sl < nslices; ++sl) {
            if (!group_is_sell[sl / (kGroupRows / kSliceRows)] || (sl + 1) * kSliceRows > nrows || slice_len[sl] == 0) continue;
            const uint32_t r0 = sl * kSliceRows, len = slice_len[sl];
            const bool uni = cls[sl] != 0;
            if (uni) {
                cur.assign(len, 0);
                for (uint32_t k = 0; k < len; ++k) cur[k] = (int)((int64_t)diag->col[diag->ptr[r0] + k] - (int64_t)r0);
            }
            if (!uni) {
                // masked slice (SellDev::mbase): the rows are sub-sequences of one ascending list of <= 16 (distance, value) pairs
                if (!want_constant || !want_masked) continue;
                std::map<int, long long> un;                                      // distance -> value bits
                bool ok = true;
                for (uint32_t l = 0; l < kSliceRows && ok; ++l) {
                    const uint32_t p0 = diag->ptr[r0 + l], p1 = diag->ptr[r0 + l + 1];
                    ok = p1 > p0 && p1 - p0 <= 16u;
                    for (uint32_t j = p0; j < p1 && ok; ++j) {
                        if (j > p0 && diag->col[j] <= diag->col[j - 1]) { ok = false; break; }      // ascending columns
                        const int d = (int)((int64_t)diag->col[j] - (int64_t)(r0 + l));
                        long long b; memcpy(&b, diag->val + j, 8);
                        auto f = un.find(d);
                        if (f == un.end()) un.emplace(d, b); else ok = f->second == b;
                    }
                    ok = ok && un.size() <= 16u;
                }
                if (!ok) continue;
                const uint32_t ulen = (uint32_t)un.size();
                cur.clear(); vkey.clear();
                std::vector<double> uv_list;
                for (auto &kv : un) { cur.push_back(kv.first); double v; memcpy(&v, &kv.second, 8); uv_list.push_back(v); }
                vkey.assign(cur.begin(), cur.end());
                for (auto &kv : un) { vkey.push_back((int)(kv.second & 0xFFFFFFFF)); vkey.push_back((int)(kv.second >> 32)); }
                auto it = lists.find(cur);
                if (it == lists.end()) {
                    if (uoff.size() + ulen + 32 > (1u << 24)) continue;
                    it = lists.emplace(cur, (uint32_t)uoff.size()).first;
                    uoff.insert(uoff.end(), cur.begin(), cur.end());
                    uoff.resize((uoff.size() + 7) / 8 * 8 + 16, 0);
                }
                auto vt = vlists.find(vkey);
                if (vt == vlists.end()) {
                    if (uval.size() + ulen + 32 > (1u << 22)) continue;
                    vt = vlists.emplace(vkey, (uint32_t)uval.size()).first;
                    uval.insert(uval.end(), uv_list.begin(), uv_list.end());
                    uval.resize((uval.size() + 7) / 8 * 8 + 16, 0.0);
                }
                if (vbase.empty()) vbase.assign(nslices, 0xFFFFFFFFu);
                if (mbase.empty()) mbase.assign(nslices, 0xFFFFFFFFu);
                ubase[sl] = it->second; vbase[sl] = vt->second;
                mbase[sl] = (ulen << 26) | (uint32_t)(rmask.size() / kSliceRows);
                for (uint32_t l = 0; l < kSliceRows; ++l) {
                    unsigned m = 0;
                    for (uint32_t j = diag->ptr[r0 + l]; j < diag->ptr[r0 + l + 1]; ++j) {
                        const int d = (int)((int64_t)diag->col[j] - (int64_t)(r0 + l));
                        m |= 1u << (unsigned)std::distance(un.begin(), un.find(d));
                    }
                    rmask.push_back((unsigned short)m);
                }
                uniform_entries += (uint64_t)len * kSliceRows; constant_entries += (uint64_t)len * kSliceRows;     // (padded entries the product no longer reads)
                masked_rows += kSliceRows;
                continue;
            }
            auto it = lists.find(cur);
            if (it == lists.end()) {
                if (uoff.size() + len + 32 > (1u << 24)) continue;            // the table stays small (scalar cache)
                it = lists.emplace(cur, (uint32_t)uoff.size()).first;
                uoff.insert(uoff.end(), cur.begin(), cur.end());
                uoff.resize((uoff.size() + 7) / 8 * 8 + 16, 0);               // batches of up to 16 entries read past the list
            }
            ubase[sl] = it->second;
            uniform_entries += (uint64_t)len * kSliceRows;
            // constant slice: entry k holds the same value in all 64 rows (SellDev::vbase)
            if (!want_constant) continue;
            const double *v0 = diag->val + diag->ptr[r0];
            if (cls[sl] != 2) continue;
            vkey.assign(cur.begin(), cur.end());                                  // distances, then the value bits
            for (uint32_t k = 0; k < len; ++k) { long long b; memcpy(&b, v0 + k, 8); vkey.push_back((int)(b & 0xFFFFFFFF)); vkey.push_back((int)(b >> 32)); }
            auto vt = vlists.find(vkey);
            if (vt == vlists.end()) {
                if (uval.size() + len + 32 > (1u << 22)) continue;
                vt = vlists.emplace(vkey, (uint32_t)uval.size()).first;
                uval.insert(uval.end(), v0, v0 + len);
                uval.resize((uval.size() + 7) / 8 * 8 + 16, 0.0);
            }
            if (vbase.empty()) vbase.assign(nslices, 0xFFFFFFFFu);
            vbase[sl] = vt->second;
            constant_entries += (uint64_t)len * kSliceRows;
        }
        if (uniform_entries == 0) { ubase.clear(); uoff.clear(); }
    }
    c->uniform_entries = uniform_entries;
    c->constant_entries = constant_entries;
    c->masked_rows = masked_rows;

    plan_mark("uniform / constant / masked slices");
    // CSR row blocks over the maximal runs of non-SELL groups
    std::vector<uint32_t> rb(nrows + 1);
    for (uint32_t g = 0; g < ngroups;) {
        if (group_is_sell[g]) { ++g; continue; }
        uint32_t g1 = g;
        while (g1 < ngroups && !group_is_sell[g1]) ++g1;
        const uint32_t r0 = g * kGroupRows, r1 = std::min(nrows, g1 * kGroupRows);
        // bicg_row_blocks works on a ptr array that starts at the run's first row
        const uint32_t nb = c->rowsplit ? bicg_row_blocks(diag->ptr + r0, r1 - r0, 8192, 256, rb.data())
                                        : bicg_row_blocks(diag->ptr + r0, r1 - r0, kRowBlockNnz, 1024, rb.data());
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t a0 = r0 + rb[b], a1 = r0 + rb[b + 1];
            const bool touches_halo = P > 1 && optr[a1] > optr[a0];
            (touches_halo ? bbnd : bint).push_back(make_uint4(a0, a1, diag->ptr[a0], diag->ptr[a1]));
        }
        g = g1;
    }
    c->n_int = (uint32_t)bint.size(); c->n_bnd = (uint32_t)bbnd.size();
    c->nblk = c->n_int + c->n_bnd;
    c->ng_int = (uint32_t)gl_int.size(); c->ng_bnd = (uint32_t)gl_bnd.size();
    c->glist_int_identity = c->ng_int == ngroups;     // every group, in order: index directly
    c->glist_all = c->ng_int + c->ng_bnd == ngroups;

    plan_mark("row blocks");
    // ---- upload
    // Only what some kernel reads goes to the GPU: the CSR val/col arrays when there are row blocks for the
    // CSR kernel (none for banded matrices: everything is on the sliced-ELL path), the 32-bit sliced-ELL
    // columns when the 16-bit offsets do not apply. (Round 1 kept all of them: 2.3 x the matrix.)
    const bool need_csr = c->nblk > 0;
    bool csr16 = c->rowsplit && need_csr && !plan_off("col16");
    std::vector<short> dcol16;
    if (csr16) {                  // rows-over-lanes kernel: 16-bit column offsets in CSR order when every entry fits
        dcol16.resize((size_t)c->nnz_d + kPadEntries, 0);
        for (uint32_t r = 0; csr16 && r < nrows; ++r)
            for (uint32_t j = diag->ptr[r]; j < diag->ptr[r + 1]; ++j) {
                const int64_t dlt = (int64_t)diag->col[j] - (int64_t)r;
                if (dlt < -32767 || dlt > 32767) { csr16 = false; break; }
                dcol16[j] = (short)dlt;
            }
    }
    c->d_val = dev_upload_padded(diag->val, need_csr ? c->nnz_d : 0, kPadEntries);
    c->d_col = dev_upload_padded(diag->col, need_csr && !csr16 ? c->nnz_d : 0, kPadEntries);
    if (csr16) c->d_col16 = dev_upload(dcol16.data(), dcol16.size());
    c->d_ptr = dev_upload(diag->ptr, (size_t)c->n_loc + 1);
    c->o_val = dev_upload(oval.data(), c->nnz_o);
    c->o_col = dev_upload(ocol.data(), c->nnz_o);
    c->o_ptr = dev_upload(optr.data(), (size_t)c->n_loc + 1);
    c->desc_int = dev_upload(bint.data(), bint.size());
    c->desc_bnd = dev_upload(bbnd.data(), bbnd.size());
    // (jagged slices: lanes whose row has ended read up to one entry past the last -- kPadEntries of slack)
    c->s_val = dev_upload_padded(sval, (size_t)sell_entries, kPadEntries);
    c->s_col = dev_upload_padded(scol, c16 ? 0 : (size_t)sell_entries, kPadEntries);
    c->matrix_bytes = (uint64_t)sell_entries * (c16 ? 10 : 12) - uniform_entries * (c16 ? 2 : 4) - constant_entries * 8ull + 2ull * masked_rows + 8ull * nslices + 4ull * (nrows + 1) +
                      (uint64_t)(c->nnz_d - c->sell_nnz) * (csr16 ? 10 : 12) + (uint64_t)c->nnz_o * 12;
    if (!vbase.empty()) {
        c->s_vbase = dev_upload(vbase.data(), vbase.size());
        c->s_uval = dev_upload(uval.data(), uval.size());
    }
    if (!mbase.empty()) {
        c->s_mbase = dev_upload(mbase.data(), mbase.size());
        c->s_rmask = dev_upload(rmask.data(), rmask.size());
    }
    if (!ubase.empty()) {
        c->s_ubase = dev_upload(ubase.data(), ubase.size());
        c->s_uoff = dev_upload(uoff.data(), uoff.size());
    }
    build_slice_desc(c, nslices, nrows, slice_len.data(), ubase, vbase, mbase, uoff, uval, rmask.empty() ? nullptr : rmask.data());
    if (P > 1) {
        // Across ranks (a z-slab of BASELINE.json configs[3]: 64 planes of 512^2 per GPU) the plane-marching product takes the
        // planes without halo entries; the halo-touching planes go through the slice-by-slice kernel behind the exchange. That
        // needs the two sets to BE whole planes: every group of a plane with a halo-touching group is halo-touching, and the
        // others form one range of planes.
        const uint32_t gpp = (c->st.on && c->st.sz % kGroupRows == 0) ? c->st.sz / kGroupRows : 0u;
        std::vector<char> bnd_plane(c->st.on ? c->st.nz : 1u, 0);
        bool ok = gpp > 0 && c->glist_all && c->nblk == 0;
        for (uint32_t g : gl_bnd) if (ok) bnd_plane[(size_t)g / gpp] = 1;
        uint32_t nb = 0, lo = c->st.nz, hi = 0;
        for (uint32_t z = 0; ok && z < c->st.nz; ++z) { if (bnd_plane[z]) ++nb; else { lo = std::min(lo, z); hi = std::max(hi, z + 1); } }
        ok = ok && (uint64_t)nb * gpp == gl_bnd.size() && lo < hi;
        for (uint32_t z = lo; ok && z < hi; ++z) ok = !bnd_plane[z];
        ok = all_ranks(comm, ok);       // (every rank takes the same form of the exchange: collective)
        if (ok) { c->st.z_lo = lo; c->st.z_hi = hi; c->st_multi = true; }
        if (getenv("BICG_PLAN_TRACE") && c->rank == 0)
            fprintf(stderr, "bicgstab_hip: plane-marching product across ranks: %s (planes %u .. %u of %u without halo entries)\n", ok ? "yes" : "no", lo, hi, c->st.nz);
    }
    c->device_matrix_bytes = (need_csr ? (csr16 ? 10ull : 12ull) * c->nnz_d : 0ull) + 4ull * (c->n_loc + 1) + 12ull * c->nnz_o + 4ull * (c->n_loc + 1) +
                             8ull * sell_entries + (c16 ? 2ull * n16 : 4ull * sell_entries) + 12ull * nslices;
    if (c16) {
        c->s_col16 = dev_upload_padded(scol16, n16_alloc, kPadEntries);
        c->s_base16 = dev_upload(slice_base16.data(), slice_base16.size());
    }
    if (win) {
        c->win_ptr = dev_upload(win_ptr.data(), win_ptr.size());
        c->win_runs = dev_upload(win_runs.data(), win_runs.size());
        c->win_slots = win_slots;
        for (uint32_t g = 0; g < ngroups; ++g) c->win_max_runs = std::max(c->win_max_runs, win_ptr[g + 1] - win_ptr[g]);
        c->win_near16 = true;
        for (uint32_t g = 0; g < ngroups && c->win_near16; ++g)
            for (uint32_t r = win_ptr[g]; r < win_ptr[g + 1]; ++r) {
                const long lo = (long)win_runs[r].x - (long)(g * kGroupRows), hi = lo + (long)(win_runs[r].y & 0xFFFFu) - 1;
                if (lo < -32767 || hi > 32767) { c->win_near16 = false; break; }
            }
        if (win_list_mode) {
            // the runs spelled out, 16 bits per column (distance from the group's first row), two slots per word: word j of thread t
            // (at lptr[g] + 256 j + t) holds slots t + 512 j (low half) and t + 512 j + 256 -- the slots thread t stages
            std::vector<uint32_t> lptr(ngroups + 1, 0u);
            std::vector<uint32_t> total(ngroups, 0u);
            for (uint32_t g = 0; g < ngroups; ++g) {
                uint32_t n = 0;
                for (uint32_t r = win_ptr[g]; r < win_ptr[g + 1]; ++r) n += win_runs[r].y & 0xFFFFu;
                total[g] = n;
                lptr[g + 1] = lptr[g] + kGroupRows * ((n + 2u * kGroupRows - 1u) / (2u * kGroupRows));
            }
            std::vector<uint32_t> list((size_t)lptr[ngroups] + 8, 0u);
            parallel_ranges(ngroups, 64, [&](size_t ga, size_t gb, int) {
                for (uint32_t g = (uint32_t)ga; g < (uint32_t)gb; ++g) {
                    uint32_t s = 0;
                    for (uint32_t r = win_ptr[g]; r < win_ptr[g + 1]; ++r)
                        for (uint32_t k = 0; k < (win_runs[r].y & 0xFFFFu); ++k, ++s) {
                            const uint32_t d = (uint32_t)((int)(win_runs[r].x + k) - (int)(g * kGroupRows)) & 0xFFFFu;
                            const uint32_t j = s / (2u * kGroupRows), rest = s % (2u * kGroupRows);
                            uint32_t &w = list[(size_t)lptr[g] + (size_t)j * kGroupRows + rest % kGroupRows];
                            w |= rest < kGroupRows ? d : d << 16;
                        }
                }
            });
            c->win_list = dev_upload(list.data(), list.size());
            c->win_lptr = dev_upload(lptr.data(), lptr.size());
            c->win_ltotal = dev_upload(total.data(), total.size());
            c->device_matrix_bytes += 4ull * list.size() + 8ull * lptr.size();
            c->matrix_bytes += 4ull * list.size() + 8ull * lptr.size();
        }
        if (!perm.empty()) c->sell_perm = dev_upload(perm.data(), perm.size());
        c->device_matrix_bytes += 4ull * win_ptr.size() + 8ull * win_runs.size();
        if (!win_list_mode) c->matrix_bytes += 4ull * win_ptr.size() + 8ull * win_runs.size();      // (the list-driven product reads the list, not the runs)
    }
    if (jag && sell_entries > 0) {
    // (with or without a window: the three-trip products of bicg_jagw.hip read one word per lane instead of two row pointers)
    // SellDev::lane_info: row in the group + its length per lane, in the order the lanes work (perm or natural)
    {
        std::vector<unsigned short> li((size_t)ngroups * kGroupRows, 0);
        std::vector<char> too_long((size_t)plan_threads(), 0);
        std::vector<uint32_t> tail_most((size_t)plan_threads(), 0u);      // entries behind the 16th of its rows, per slice (k_spmm_jpipe keeps them in LDS)
        parallel_ranges(ngroups, 64, [&](size_t ga, size_t gb, int part) {
            for (uint32_t g = (uint32_t)ga; g < (uint32_t)gb; ++g) {
                uint32_t tail = 0;
                for (uint32_t t = 0; t < kGroupRows; ++t) {
                    const uint32_t in_group = perm.empty() ? t : perm[(size_t)g * kGroupRows + t], r = g * kGroupRows + in_group;
                    const uint32_t n = (r < nrows && group_is_sell[g]) ? diag->ptr[r + 1] - diag->ptr[r] : 0u;
                    if (n > 255u) too_long[(size_t)part] = 1;
                    li[(size_t)g * kGroupRows + t] = (unsigned short)(in_group | (n << 8));
                    if (t % kSliceRows == 0) tail = 0;
                    tail += n > 16u ? n - 16u : 0u;
                    tail_most[(size_t)part] = std::max(tail_most[(size_t)part], tail);
                }
            }
        });
        bool ok = true;
        for (char b : too_long) ok = ok && !b;
        for (uint32_t t : tail_most) c->jag_tail16_max = std::max(c->jag_tail16_max, t);
        if (ok) {
            c->lane_info = dev_upload(li.data(), li.size());
            c->matrix_bytes += 2ull * li.size();
            c->device_matrix_bytes += 2ull * li.size();
            // (one rank: every product goes through the three-trip kernels, which do not read the row pointers)
            if (P == 1 && c->matrix_bytes > 4ull * (nrows + 1)) c->matrix_bytes -= 4ull * (nrows + 1);
        }
        if (const char *v = plan_tok("jagw")) c->jagw_fast = atoi(v) != 0;
    }
    }
    c->s_base = dev_upload(slice_base.data(), slice_base.size());
    c->s_len = dev_upload(slice_len.data(), slice_len.size());
    c->glist_int = dev_upload(gl_int.data(), gl_int.size());
    c->glist_bnd = dev_upload(gl_bnd.data(), gl_bnd.size());
    c->send_idx = dev_upload(send_idx.data(), c->nsend);
    c->sendbuf = dev_alloc<double>(c->nsend);

    plan_mark("upload");
    // ---- peer-to-peer transport: publish this rank's halo landing ring, learn where every entry
    // of the send list lands in the ring of the rank that needs it (collective)
    c->p2p = comm->p2p;
    std::vector<unsigned long long> dst0, dstride;
    if (const char *sv = test_tok("p2p-fault-after")) c->fault_after = atoi(sv);
    // in-kernel collect needs the HEAVY kernel instantiations (occupancy 5 instead of 8 waves per SIMD,
    // ~3 % per SpMV): worth it unless the local problem is so large that 3 % exceeds the ~10 us per
    // iteration the separate apply kernels cost
    c->inline_apply = c->nnz_d < 40000000u;
    if (const char *sv = knob_x("BICG_P2P_INLINE_APPLY")) c->inline_apply = atoi(sv) != 0;
    if (c->p2p && !c->single()) {
        c->halo_ring = (llword *)c->p2p->alloc(sizeof(llword) * 2 * (size_t)kHaloRing * c->halo);
        std::vector<void *> rings;
        if (c->p2p->share(c->halo_ring, rings, c->ring_mapped) != 0)
            die("bicg_create", "could not map the halo rings of the other ranks (peer-to-peer transport)");
        // to rank p: where ITS values land in my ring, and my ring's slot size
        std::vector<int> mine(2 * (size_t)P), theirs(2 * (size_t)P, 0), cnt(P, 2 * (int)sizeof(int)), dsp(P);
        for (int p = 0; p < P; ++p) {
            mine[2 * p] = c->rdsp[p]; mine[2 * p + 1] = (int)c->halo;
            dsp[p] = 2 * p * (int)sizeof(int);
        }
        comm->alltoallv_host(mine.data(), cnt.data(), dsp.data(), theirs.data(), cnt.data(), dsp.data());
        dst0.assign(c->nsend ? c->nsend : 1, 0ull); dstride.assign(c->nsend ? c->nsend : 1, 0ull);
        for (int p = 0; p < P; ++p)
            for (int j = 0; j < c->scnt[p]; ++j) {
                const size_t i = (size_t)c->sdsp[p] + j;
                dst0[i] = (unsigned long long)(uintptr_t)rings[p] + 16ull * ((unsigned long long)theirs[2 * p] + j);
                dstride[i] = 16ull * (unsigned long long)theirs[2 * p + 1];
            }
        c->push_dst0 = dev_upload(dst0.data(), dst0.size());
        c->push_stride = dev_upload(dstride.data(), dstride.size());
        c->ll_fused = c->n_bnd == 0 && c->ng_int + c->ng_bnd > 0;
        // Ragged rows (jagged slices): the launch with the exchange inside runs k_spmv_sell's loop over EVERY group, the rank's
        // halo-free groups included; as separate launches -- push, interior, unpack, boundary -- the interior goes through the
        // three-trip products of bicg_jagw.hip. Worth two more launches when the interior is large (measured with two 800 k-row
        // ranks of the RCM-numbered mesh matrix sharing a GPU: profiles/r06/ragged_ranks_fused_or_split.txt); BICG_PLAN="halo-fused=0|1" decides.
        if (c->ll_fused && c->sell_jag && c->lane_info && c->jagw_fast && (uint64_t)c->nnz_d >= 4000000ull) c->ll_fused = false;
        if (const char *sv = plan_tok("halo-fused")) c->ll_fused = c->n_bnd == 0 && c->ng_int + c->ng_bnd > 0 && atoi(sv) != 0;
        if (const char *sv = knob_x("BICG_P2P_FUSED")) c->ll_fused = c->ll_fused && atoi(sv) != 0;
        if (c->ll_fused) {
            std::vector<uint32_t> order(gl_int);
            order.insert(order.end(), gl_bnd.begin(), gl_bnd.end());
            c->glist_ll = dev_upload(order.data(), order.size());
        }
    } else {
        c->p2p = nullptr;
    }

    ctx_state(c, comm, ngroups);
    if (const char *sv = test_tok("spin-ticks")) c->spin_ticks = strtoull(sv, nullptr, 10);
    // Round 4: with the products alternating direction and reading no column index in uniform slices, a big block is faster
    // as two plain products + two element-wise kernels (Transport-shaped, one GPU: 139.0 vs 152.0 us per pipelined iteration;
    // profiles/NOTES.md): the fused two-launch form stays what it was built for -- the latency-bound ranks.
    c->fuse_pipe = c->fuse_small;
    if (const char *sv = plan_tok("fuse-pipe")) c->fuse_pipe = atoi(sv) != 0;
    else if (const char *pv = plan_tok("pipe-probe")) c->pipe_probe = atoi(pv);
    c->spmm_ok = all_ranks(comm, spmm_possible(c));
    c->fuse_plan_ok = all_ranks(comm, c->glist_all && c->nblk == 0 && (c->single() || (c->p2p && c->ll_fused)));
    BICG_HIP(hipHostMalloc((void **)&c->hS, sizeof(Scal), hipHostMallocDefault));
    memset(c->hS, 0, sizeof(Scal));
    {   // persistent pipelined iteration for latency-bound ranks: available when the plan fits on EVERY rank
        const bool off = knob_tok("BICG_PERSIST", "0") || knob_tok("BICG_PERSIST", "off");
        bool mine = !off && persist_build(c, diag, optr, ocol, oval, send_idx, dst0, dstride);
        c->persist_on = all_ranks(comm, mine);
        if (const char *pp = knob_x("BICG_PERSIST_PLAIN")) c->persist_plain = atoi(pp) != 0;
        if (!c->persist_on && mine) { for (void *p : c->persist_mem) (void)hipFree(p); c->persist_mem.clear(); c->persist = PersistArgs{}; }
    }

    plan_mark("transport, persistent plan");
    ctx_streams(c, P);
    plan_mark("streams");
    preload_for(c);
    plan_mark("code objects");
    return c;
}

// Single rank, the matrix ALREADY in device memory as CSR: the sliced-ELL plan (slice lengths, bases, the column-major
// padded copy, 16-bit column offsets when they fit) is built by kernels (bicg_plan_device.hip) -- no host copy of the
// matrix ever exists. This is what makes BASELINE.json configs[3] at its stated size fit a bench run: the 512^3 Laplacian
// (134 M rows, 938 M non-zeros, 11 GB of CSR) is generated on the GPU (bicg_stencil7_device) and planned in a fraction of
// a second, where the one-thread host plan of bicg_create would take the better part of a minute after a 15 GB transfer.
// Blocks whose rows are too ragged for padded slices (or long enough for the rows-over-lanes kernel) are refused: the
// caller downloads the CSR and takes bicg_create.
bicg_ctx *bicg_create_device_csr(const double *val_d, const unsigned int *col_d, const unsigned int *ptr_d, unsigned int rows,
                                 double *plan_seconds)
{
    Comm *comm = comm_get();
    BICG_HIP(hipSetDevice(comm->device));
    if (comm->nranks != 1) { fprintf(stderr, "ERROR: bicg_create_device_csr: single rank only\n"); return nullptr; }
    warn_unknown_switches();
    if (rows == 0) { fprintf(stderr, "ERROR: bicg_create_device_csr: empty matrix\n"); return nullptr; }
    const double t0 = now_sec();
    unsigned nnz = 0;
    BICG_HIP(hipMemcpy(&nnz, ptr_d + rows, sizeof(unsigned), hipMemcpyDeviceToHost));
    const uint32_t nslices = (rows + kSliceRows - 1) / kSliceRows, ngroups = (rows + kGroupRows - 1) / kGroupRows;
    uint32_t *slen_d = dev_alloc<uint32_t>(nslices);
    int *far_d = dev_alloc<int>(1);
    BICG_HIP(hipMemset(slen_d, 0, sizeof(uint32_t) * nslices));
    BICG_HIP(hipMemset(far_d, 0, sizeof(int)));
    launch_plan_rowstats(ptr_d, col_d, rows, slen_d, far_d, nullptr);
    std::vector<uint32_t> slen(nslices), sbase(nslices), sbase16(nslices);
    int far = 0;
    BICG_HIP(hipMemcpy(slen.data(), slen_d, sizeof(uint32_t) * nslices, hipMemcpyDeviceToHost));
    BICG_HIP(hipMemcpy(&far, far_d, sizeof(int), hipMemcpyDeviceToHost));
    uint64_t entries = 0, n16 = 0, padded_rows = 0;
    uint32_t longest = 0;
    for (uint32_t sl = 0; sl < nslices; ++sl) {
        sbase[sl] = (uint32_t)entries; sbase16[sl] = (uint32_t)n16;
        entries += (uint64_t)slen[sl] * kSliceRows;
        n16 += (uint64_t)((slen[sl] + 3) / 4) * 4 * kSliceRows;
        padded_rows += (uint64_t)slen[sl] * std::min<uint32_t>(kSliceRows, rows - sl * kSliceRows);
        longest = std::max(longest, slen[sl]);
    }
    const bool c16 = !far && n16 < 0xFFFFFF00ull && !plan_off("col16");
    const char *why = nullptr;
    if (entries >= 0xFFFFFF00ull) why = "more than 2^32 sliced-ELL entries";
    else if (padded_rows > (uint64_t)nnz + nnz / 50) why = "ragged rows (jagged slices are planned on the host)";
    else if ((uint64_t)nnz / rows >= 128 && ngroups < 512) why = "long rows (the rows-over-lanes plan is built on the host)";
    else if (longest > std::max<uint64_t>(64, 4 * (uint64_t)nnz / rows)) why = "a row much longer than the average";
    if (why) {
        fprintf(stderr, "bicgstab_hip: bicg_create_device_csr: %s -- use bicg_create\n", why);
        BICG_HIP(hipFree(slen_d)); BICG_HIP(hipFree(far_d));
        return nullptr;
    }
    bicg_ctx *c = new bicg_ctx;
    c->comm = comm; c->device = comm->device; c->nranks = 1; c->rank = 0;
    g_live.push_back(c);
    c->n_loc = rows; c->n_glob = rows; c->nnz_d = nnz;
    if (const char *sv = knob_x("BICG_SELL_NT")) c->sell_nt_env = atoi(sv);
    if (const char *sv = knob_x("BICG_SELL_ALT")) c->sell_alt = atoi(sv);
    if (const char *sv = knob_x("BICG_SELL_XCD")) c->sell_xcd = atoi(sv);
    if (const char *sv = test_tok("force-comm")) c->force_comm = atoi(sv) != 0;
    if (c->force_comm) die("bicg_create_device_csr", "BICG_TEST=force-comm is not supported on this path");
    c->overlap = nnz >= 6000000u; c->fuse_small = nnz < 6000000u;
    c->scnt.assign(1, 0); c->sdsp.assign(1, 0); c->rcnt.assign(1, 0); c->rdsp.assign(1, 0);
    c->sell_entries = entries; c->sell_nnz = nnz; c->sell_rows = rows; c->sell_jag = false;
    c->s_val = dev_alloc<double>((size_t)entries + kPadEntries);
    BICG_HIP(hipMemset(c->s_val, 0, sizeof(double) * ((size_t)entries + kPadEntries)));
    if (c16) {
        c->s_col16 = dev_alloc<short>((size_t)n16 + kPadEntries);
        BICG_HIP(hipMemset(c->s_col16, 0, sizeof(short) * ((size_t)n16 + kPadEntries)));
        c->s_base16 = dev_upload(sbase16.data(), sbase16.size());
        c->s_col = dev_alloc<uint32_t>(kPadEntries);
    } else {
        c->s_col = dev_alloc<uint32_t>((size_t)entries + kPadEntries);
        BICG_HIP(hipMemset(c->s_col, 0, sizeof(uint32_t) * ((size_t)entries + kPadEntries)));
    }
    c->s_base = dev_upload(sbase.data(), sbase.size());
    c->s_len = slen_d;
    launch_plan_fill(ptr_d, col_d, val_d, rows, c->s_base, c->s_base16, c->s_val, c16 ? nullptr : c->s_col, c16 ? c->s_col16 : nullptr, nullptr);
    // uniform slices (SellDev::ubase): found by a kernel, grouped by the hash of their distance lists here; one list per group
    // is fetched from the CSR (a stencil has a few dozen)
    uint64_t uniform_entries = 0, constant_entries = 0;
    uint32_t far_rows = 0;
    if (!plan_off("uniform")) {
        const bool want_constant = !plan_off("constant");
        unsigned long long *uh_d = dev_alloc<unsigned long long>(2 * (size_t)nslices), *vh_d = uh_d + nslices;
        BICG_HIP(hipMemset(uh_d, 0, sizeof(unsigned long long) * 2 * (size_t)nslices));
        launch_plan_uniform(ptr_d, col_d, val_d, rows, uh_d, want_constant ? vh_d : nullptr, nullptr);
        std::vector<unsigned long long> uh(nslices), vh(nslices);
        BICG_HIP(hipMemcpy(uh.data(), uh_d, sizeof(unsigned long long) * nslices, hipMemcpyDeviceToHost));
        BICG_HIP(hipMemcpy(vh.data(), vh_d, sizeof(unsigned long long) * nslices, hipMemcpyDeviceToHost));
        BICG_HIP(hipFree(uh_d));
        // tests: every hash lands in one of TWO buckets -- slices with different lists collide in their thousands and
        // k_plan_verify has to catch each one (tests/test_full_size.py::test_device_plan_survives_hash_collisions)
        const bool collide = test_tok("plan-collide") != nullptr;
        if (collide) for (uint32_t sl = 0; sl < nslices; ++sl) { if (uh[sl]) uh[sl] = 1ull + (uh[sl] >> 63); if (vh[sl]) vh[sl] = 1ull + (vh[sl] >> 63); }
        std::vector<uint32_t> vbase, mbase;
        std::vector<double> uval, vals;
        std::map<unsigned long long, uint32_t> vlists;
        std::vector<uint32_t> ubase(nslices, 0xFFFFFFFFu);
        std::vector<int> uoff;
        std::map<unsigned long long, uint32_t> lists;
        std::vector<uint32_t> cols;
        for (uint32_t sl = 0; sl < nslices; ++sl) {
            if (!uh[sl]) continue;
            auto it = lists.find(uh[sl]);
            if (it == lists.end()) {
                if (lists.size() >= 4096) continue;                           // not a structured matrix: leave the rest to col / col16
                const uint32_t r0 = sl * kSliceRows, len = slen[sl];
                uint32_t p0 = 0;
                BICG_HIP(hipMemcpy(&p0, ptr_d + r0, sizeof(uint32_t), hipMemcpyDeviceToHost));
                cols.resize(len);
                BICG_HIP(hipMemcpy(cols.data(), col_d + p0, sizeof(uint32_t) * len, hipMemcpyDeviceToHost));
                it = lists.emplace(uh[sl], (uint32_t)uoff.size()).first;
                for (uint32_t k = 0; k < len; ++k) uoff.push_back((int)((int64_t)cols[k] - (int64_t)r0));
                uoff.resize((uoff.size() + 7) / 8 * 8 + 16, 0);
            }
            ubase[sl] = it->second;
            uniform_entries += (uint64_t)slen[sl] * kSliceRows;
            if (!vh[sl]) continue;                                                // constant slice (SellDev::vbase)
            auto vt = vlists.find(vh[sl]);
            if (vt == vlists.end()) {
                if (vlists.size() >= 4096) continue;
                const uint32_t r0 = sl * kSliceRows, len = slen[sl];
                uint32_t p0 = 0;
                BICG_HIP(hipMemcpy(&p0, ptr_d + r0, sizeof(uint32_t), hipMemcpyDeviceToHost));
                vals.resize(len);
                BICG_HIP(hipMemcpy(vals.data(), val_d + p0, sizeof(double) * len, hipMemcpyDeviceToHost));
                vt = vlists.emplace(vh[sl], (uint32_t)uval.size()).first;
                uval.insert(uval.end(), vals.begin(), vals.end());
                uval.resize((uval.size() + 7) / 8 * 8 + 16, 0.0);
            }
            if (vbase.empty()) vbase.assign(nslices, 0xFFFFFFFFu);
            vbase[sl] = vt->second;
            constant_entries += (uint64_t)slen[sl] * kSliceRows;
        }
        for (int d : uoff) far_rows = std::max<uint32_t>(far_rows, (uint32_t)std::abs(d));      // the farthest distance of a uniform slice
        // masked slices (SellDev::mbase): the slices next to a grid face. Found by a kernel (hash of the slice's list of
        // (distance, value) pairs), one representative per hash is fetched and its list rebuilt here, the rows' masks are
        // written by a second pass over the slices that were kept.
        if (want_constant && !plan_off("masked")) {
            unsigned long long *mh_d = dev_alloc<unsigned long long>(nslices);
            BICG_HIP(hipMemset(mh_d, 0, sizeof(unsigned long long) * nslices));
            launch_plan_masked(ptr_d, col_d, val_d, rows, mh_d, nullptr, nullptr, nullptr);
            std::vector<unsigned long long> mh(nslices);
            BICG_HIP(hipMemcpy(mh.data(), mh_d, sizeof(unsigned long long) * nslices, hipMemcpyDeviceToHost));
            BICG_HIP(hipFree(mh_d));
            if (collide) for (uint32_t sl = 0; sl < nslices; ++sl) if (mh[sl]) mh[sl] = (mh[sl] & 31ull) | (32ull << (mh[sl] >> 63));
            std::map<unsigned long long, std::pair<uint32_t, uint32_t>> mlists;       // hash -> (position in uoff, position in uval)
            std::vector<uint32_t> rp(kSliceRows + 1), rc;
            std::vector<double> rv;
            uint32_t nmasked = 0;
            for (uint32_t sl = 0; sl < nslices; ++sl) {
                if (ubase[sl] != 0xFFFFFFFFu || !mh[sl] || (sl + 1) * kSliceRows > rows) continue;
                const uint32_t ulen = (uint32_t)(mh[sl] & 31ull);
                auto it = mlists.find(mh[sl]);
                if (it == mlists.end()) {
                    if (mlists.size() >= 4096) continue;
                    const uint32_t r0 = sl * kSliceRows;
                    BICG_HIP(hipMemcpy(rp.data(), ptr_d + r0, sizeof(uint32_t) * (kSliceRows + 1), hipMemcpyDeviceToHost));
                    const uint32_t ne = rp[kSliceRows] - rp[0];
                    rc.resize(ne); rv.resize(ne);
                    BICG_HIP(hipMemcpy(rc.data(), col_d + rp[0], sizeof(uint32_t) * ne, hipMemcpyDeviceToHost));
                    BICG_HIP(hipMemcpy(rv.data(), val_d + rp[0], sizeof(double) * ne, hipMemcpyDeviceToHost));
                    std::map<int, double> un;
                    for (uint32_t l = 0; l < kSliceRows; ++l)
                        for (uint32_t j = rp[l]; j < rp[l + 1]; ++j) un.emplace((int)((int64_t)rc[j - rp[0]] - (int64_t)(r0 + l)), rv[j - rp[0]]);
                    if (un.size() != ulen) continue;                              // (cannot happen: the kernel built the same list)
                    it = mlists.emplace(mh[sl], std::make_pair((uint32_t)uoff.size(), (uint32_t)uval.size())).first;
                    for (auto &kv : un) { uoff.push_back(kv.first); uval.push_back(kv.second); far_rows = std::max<uint32_t>(far_rows, (uint32_t)std::abs(kv.first)); }
                    uoff.resize((uoff.size() + 7) / 8 * 8 + 16, 0);
                    uval.resize((uval.size() + 7) / 8 * 8 + 16, 0.0);
                }
                if (mbase.empty()) mbase.assign(nslices, 0xFFFFFFFFu);
                if (vbase.empty()) vbase.assign(nslices, 0xFFFFFFFFu);
                ubase[sl] = it->second.first; vbase[sl] = it->second.second;
                mbase[sl] = (ulen << 26) | nmasked++;
                uniform_entries += (uint64_t)slen[sl] * kSliceRows; constant_entries += (uint64_t)slen[sl] * kSliceRows;
            }
            if (nmasked) {
                c->s_mbase = dev_upload(mbase.data(), mbase.size());
                c->s_rmask = dev_alloc<unsigned short>((size_t)nmasked * kSliceRows);
                BICG_HIP(hipMemset(c->s_rmask, 0, sizeof(unsigned short) * (size_t)nmasked * kSliceRows));
                launch_plan_masked(ptr_d, col_d, val_d, rows, nullptr, c->s_mbase, c->s_rmask, nullptr);
                BICG_HIP(hipDeviceSynchronize());
                c->masked_rows = (uint64_t)nmasked * kSliceRows;
            }
        }
        if (uniform_entries) {
            c->s_ubase = dev_upload(ubase.data(), ubase.size());
            c->s_uoff = dev_upload(uoff.data(), uoff.size());
        }
        if (constant_entries) {
            c->s_vbase = dev_upload(vbase.data(), vbase.size());
            c->s_uval = dev_upload(uval.data(), uval.size());
        }
        // The groups above are keyed by 64-bit hashes: every list-driven slice is now compared with the list it was given
        // (k_plan_verify), and a slice that differs -- a collision -- goes back to its stored columns and values, which
        // launch_plan_fill has written for every slice. (The host plan keys on the full lists and needs no such pass.)
        if (uniform_entries) {
            unsigned char *bad_d = dev_alloc<unsigned char>(nslices);
            BICG_HIP(hipMemset(bad_d, 0, nslices));
            launch_plan_verify(ptr_d, col_d, val_d, rows, slen_d, c->s_ubase, c->s_vbase, c->s_mbase, c->s_rmask, c->s_uoff, c->s_uval, bad_d, nullptr);
            std::vector<unsigned char> bad(nslices);
            BICG_HIP(hipMemcpy(bad.data(), bad_d, nslices, hipMemcpyDeviceToHost));
            BICG_HIP(hipFree(bad_d));
            uint32_t nbad = 0;
            for (uint32_t sl = 0; sl < nslices; ++sl) {
                if (!bad[sl]) continue;
                ++nbad;
                const uint64_t e = (uint64_t)slen[sl] * kSliceRows;
                uniform_entries -= e;
                if (!vbase.empty() && vbase[sl] != 0xFFFFFFFFu) { constant_entries -= e; vbase[sl] = 0xFFFFFFFFu; }
                if (!mbase.empty() && mbase[sl] != 0xFFFFFFFFu) { c->masked_rows -= kSliceRows; mbase[sl] = 0xFFFFFFFFu; }
                ubase[sl] = 0xFFFFFFFFu;
            }
            c->plan_collisions = nbad;
            if (nbad) {
                BICG_HIP(hipMemcpy(c->s_ubase, ubase.data(), sizeof(uint32_t) * nslices, hipMemcpyHostToDevice));
                if (c->s_vbase) BICG_HIP(hipMemcpy(c->s_vbase, vbase.data(), sizeof(uint32_t) * nslices, hipMemcpyHostToDevice));
                if (c->s_mbase) BICG_HIP(hipMemcpy(c->s_mbase, mbase.data(), sizeof(uint32_t) * nslices, hipMemcpyHostToDevice));
                if (getenv("BICG_PLAN_TRACE")) fprintf(stderr, "bicgstab_hip: %u list-driven slices did not match their list (hash collision): stored as general slices\n", nbad);
            }
        }
        if (constant_entries) build_slice_desc(c, nslices, rows, slen.data(), ubase, vbase, mbase, uoff, uval, nullptr);
    }
    c->uniform_entries = uniform_entries;
    c->constant_entries = constant_entries;
    c->far_rows = far_rows;
    c->d_ptr = dev_alloc<uint32_t>((size_t)rows + 1);
    BICG_HIP(hipMemcpy(c->d_ptr, ptr_d, sizeof(uint32_t) * ((size_t)rows + 1), hipMemcpyDeviceToDevice));
    c->d_val = dev_alloc<double>(kPadEntries); c->d_col = dev_alloc<uint32_t>(kPadEntries);
    c->o_val = dev_alloc<double>(1); c->o_col = dev_alloc<uint32_t>(1);
    c->o_ptr = dev_alloc<uint32_t>((size_t)rows + 1);
    BICG_HIP(hipMemset(c->o_ptr, 0, sizeof(uint32_t) * ((size_t)rows + 1)));
    c->desc_int = dev_alloc<uint4>(1); c->desc_bnd = dev_alloc<uint4>(1);
    c->glist_int = dev_alloc<uint32_t>(1); c->glist_bnd = dev_alloc<uint32_t>(1);
    c->send_idx = dev_alloc<uint32_t>(1); c->sendbuf = dev_alloc<double>(1);
    c->ng_int = ngroups; c->ng_bnd = 0; c->n_int = c->n_bnd = c->nblk = 0;
    c->glist_int_identity = true; c->glist_all = true;
    sell_order_for_big_grids(c, ngroups);
    c->matrix_bytes = entries * (c16 ? 10 : 12) - uniform_entries * (c16 ? 2 : 4) - constant_entries * 8ull + 2ull * c->masked_rows + 8ull * nslices + 4ull * ((uint64_t)rows + 1);
    if (c->s_desc) c->matrix_bytes += 8ull * nslices;
    c->device_matrix_bytes = 8ull * ((uint64_t)rows + 1) + 8ull * entries + (c16 ? 2ull * n16 : 4ull * entries) + 12ull * nslices;
    BICG_HIP(hipFree(far_d));
    ctx_state(c, comm, ngroups);
    if (const char *sv = test_tok("spin-ticks")) c->spin_ticks = strtoull(sv, nullptr, 10);
    c->fuse_pipe = c->fuse_small;
    if (const char *sv = plan_tok("fuse-pipe")) c->fuse_pipe = atoi(sv) != 0;
    else if (const char *pv = plan_tok("pipe-probe")) c->pipe_probe = atoi(pv);
    c->spmm_ok = spmm_possible(c);
    c->fuse_plan_ok = true;
    BICG_HIP(hipHostMalloc((void **)&c->hS, sizeof(Scal), hipHostMallocDefault));
    memset(c->hS, 0, sizeof(Scal));
    ctx_streams(c, 1);
    preload_for(c);
    if (plan_seconds) *plan_seconds = now_sec() - t0;
    return c;
}

namespace {
// the halo landing ring lives in the transport's shared memory: give it back while the transport exists
void release_p2p(bicg_ctx *c)
{
    if (!c->p2p) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    c->p2p->unmap(c->ring_mapped);
    c->p2p->release(c->halo_ring);
    c->ring_mapped.clear(); c->halo_ring = nullptr; c->p2p = nullptr;
}
}  // namespace

// called by comm_set() before the communicator goes away (bicg_comm.cpp)
extern "C++" {
void bicg::contexts_orphan()
{
    for (bicg_ctx *c : g_live) { release_p2p(c); c->comm = nullptr; }
}
}

void bicg_destroy(bicg_ctx *c)
{
    if (!c) return;
    g_live.erase(std::remove(g_live.begin(), g_live.end(), c), g_live.end());
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    void *ptrs[] = {c->d_val, c->d_col, c->d_ptr, c->o_val, c->o_col, c->o_ptr, c->desc_int, c->desc_bnd, c->s_val, c->s_col, c->s_base, c->s_len, c->s_col16, c->s_base16, c->s_ubase, c->s_uoff, c->s_vbase, c->s_uval, c->s_mbase, c->s_rmask, c->s_desc, c->s_uoff8, c->st_code, c->st_tab, c->st_cmask, c->st_wbits, c->d_col16, c->win_ptr, c->win_runs, c->win_list, c->win_lptr, c->win_ltotal, c->sell_perm, c->lane_info, c->waitlog, c->sh_dev, c->sh_arrays, c->p_set, c->x_set, c->glist_int, c->glist_bnd,
                    c->send_idx, c->sendbuf, c->slab, c->partial, c->shard_tot, c->counter, c->Sbuf, c->trace, c->sw_buf,
                    c->wpart[0], c->wpart[1], c->shard_ll, c->tail_tab, c->tail_shard, c->alarm, c->mm_in, c->mm_xt, c->mm_yt, c->mm_part, c->mm_out, c->mm_sigma};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (void *p : c->persist_mem) if (p) (void)hipFree(p);
    release_p2p(c);
    if (c->push_dst0) (void)hipFree(c->push_dst0);
    if (c->push_stride) (void)hipFree(c->push_stride);
    if (c->glist_ll) (void)hipFree(c->glist_ll);
    if (c->hS) (void)hipHostFree(c->hS);
    if (c->h_alarm) (void)hipHostFree(c->h_alarm);
    for (int i = 0; i < kEvRing; ++i) {
        for (hipEvent_t e : {c->ev_pack[i], c->ev_halo[i], c->ev_dots[i], c->ev_red[i]})
            if (e) (void)hipEventDestroy(e);      // a context that failed early in bicg_create has none
    }
    for (auto &e : c->tev) (void)hipEventDestroy(e);
    for (auto &e : c->region_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->sec_ev) (void)hipEventDestroy(e);
    for (auto &ge : c->graph_exec) if (ge) (void)hipGraphExecDestroy(ge);
    if (c->sc) (void)hipStreamDestroy(c->sc);
    if (c->sm) (void)hipStreamDestroy(c->sm);
    delete c;
}


}  // extern "C"
