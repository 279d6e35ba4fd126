// bicg_stencil.hip -- the plane-marching product: y = A x for a block whose sliced-ELL plan found the 7-point stencil of a
// grid in its lists (struct StencilDev, bicg_device.h; BASELINE.json configs[3], the 512^3 Laplacian of CA-BiCGStab).
//
// Why a kernel of its own: the slice-by-slice product of bicg_kernels.hip (k_spmv_sell, the loop over list-driven slices) moves
// 2.3 GB per 512^3 product at the memory side but fills 5.4 GB into the vector L1s -- 20 cache lines per 64-row slice, every x
// line fetched by five slices at five different times -- and is bound by those fills (one CU sustains ~10 B/cycle from its L1's
// miss path: profiles/r04/laplace512_spmv_counters_lists_loop.txt). Here a wavefront keeps what it has fetched:
//   * it owns R consecutive grid lines of one 64-wide x segment and walks through zl planes; the x values of its own rows in
//     the planes z - 1, z, z + 1 are registers (the -sz / own / +sz entries), plane z + 2 is in flight;
//   * the -sy / +sy entries of a line are the registers of the line below / above, and ONE halo load per side of the R lines;
//   * the -1 / +1 entries are the neighbouring lane's own value (DPP wave shift); the two lanes at the ends of the segment get
//     theirs from one load per line that touches the two adjacent cache lines.
// Per slice that is (4 R + 8) / R + 2 fills = 8 for R = 4 instead of 20, no LDS and no barrier: the wavefronts stay
// independent (a barrier per slice is what the first attempt at sharing lines through LDS lost to, profiles/NOTES.md round 4).
//
// What is computed: row i adds val_k * x[i + d_k] over the entries its list has, in the list's order = ascending column =
// stored order, each product rounded before it is added (-ffp-contract=off), y_i = 0.0 + that sum -- bit for bit the sum of
// mult() (reference src/matrix.c:506-515) and of the other sliced-ELL products. Every x value is read at its literal distance
// from the row (the "registers of the line below" ARE x[i - sy]); which entries a row has comes from the plan (StencilTab::bits
// for the slice, StencilDev::cmask per row where rows of a slice differ), never from assumptions about grid faces.
// The dot epilogue (NDOT) and its reduction are those of k_spmv_sell; EPI = 1 adds CA-BiCGStab's q = r - alpha s, y = w - alpha z,
// (q,y), (y,y) (reference src/solver.c:225-232, FQY of bicg_kernels.hip, the same expressions) on the own rows behind z = A s:
// s_i is the register the product multiplied, z_i the sum it has just formed -- neither is read again.
#include "bicg_device.h"
#include "bicg_devfn.h"
#include "bicg_reduce.h"

#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

namespace bicg {

#define BICG_KCONST __attribute__((address_space(4)))

// lane i <- v of lane i - 1 (CTRL 0x138, wave_shr:1) or lane i + 1 (0x130, wave_shl:1); the lane without a source keeps `edge`
template <int CTRL>
__device__ __forceinline__ double st_wave_shift(double v, double edge)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(edge), lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double st_ld(const char *base, uint32_t off) { return *reinterpret_cast<const double *>(base + off); }
__device__ __forceinline__ void st_st(char *base, uint32_t off, double v, int nt = 0)
{
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<double *>(base + off));
    else *reinterpret_cast<double *>(base + off) = v;
}

// the seven products in canonical = stored order (src/matrix.c:506-515)
__device__ __forceinline__ double st_sum_all(const double (&cv)[7], const double (&xv)[7])
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += cv[k] * xv[k];
    return s;
}
__device__ __forceinline__ double st_sum_bits(const double (&cv)[7], const double (&xv)[7], uint32_t bits)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if ((bits >> k) & 1u) s += cv[k] * xv[k];       // an entry the row does not have is not added
    return s;
}

template <int R> struct st_codes;
template <> struct st_codes<2> { typedef unsigned type __attribute__((ext_vector_type(2))); };
template <> struct st_codes<4> { typedef unsigned type __attribute__((ext_vector_type(4))); };
template <> struct st_codes<8> { typedef unsigned type __attribute__((ext_vector_type(8))); };

// what a step requests one plane ahead, besides the plane itself
template <int R, int NDOT, int EPI, bool MCOL> struct StAux {
    double lo, hi;                       // x of the line below line 0 / above line R - 1
    double e[R];                         // lanes 0..31: x left of the segment, lanes 32..63: right of it
    double u[(NDOT > 0 && !EPI) ? R : 1];
    double r[EPI ? R : 1], w[EPI ? R : 1];
    uint32_t cm[MCOL ? R : 1];
};

// One wavefront's state while it marches: scalars (rows, table position, the list pair in registers) and the dot sums
template <int R, int ND> struct StWalk {
    const char *xb, *ub; char *yb, *rb, *wb;
    const unsigned char *cmask;
    const BICG_KCONST StencilTab *tab;
    const BICG_KCONST unsigned *code;       // the R codes of the current plane
    uint32_t n, sy, sz, nz, lane8, hi_half, lane;
    uint32_t rz, mz, mline, mstep;          // first row of line 0 in the current plane; the same place in cmask
    uint32_t cstep;                         // codes per plane
    uint32_t cidx, cbits;
    int nt;
    double cv[7];
    double alpha;
    double acc[ND];
};

// what the plane whose line 0 starts at row r (cmask position mr) needs besides the own rows
template <int NDOT, int R, int EPI, bool MCOL, int ND>
__device__ __forceinline__ void st_aux(const StWalk<R, ND> &w, uint32_t r, uint32_t mr, StAux<R, NDOT, EPI, MCOL> &A)
{
    // neighbours outside the vector belong to no row's list: those loads are pointed at rows that exist
    const uint32_t rlo = r >= w.sy ? r - w.sy : r, rhi = r + (uint32_t)R * w.sy < w.n ? r + (uint32_t)R * w.sy : r + (uint32_t)(R - 1) * w.sy;
    A.lo = st_ld(w.xb, (rlo << 3) + w.lane8);
    A.hi = st_ld(w.xb, (rhi << 3) + w.lane8);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t rj = r + (uint32_t)j * w.sy;
        const uint32_t left = rj > 0u ? rj - 1u : rj, right = rj + (uint32_t)kSliceRows < w.n ? rj + (uint32_t)kSliceRows : rj + (uint32_t)kSliceRows - 1u;
        A.e[j] = st_ld(w.xb, (left << 3) + (w.hi_half & ((right - left) << 3)));
        if (NDOT > 0 && !EPI) A.u[j] = st_ld(w.ub, (rj << 3) + w.lane8);
        if (EPI) { A.r[j] = st_ld(w.rb, (rj << 3) + w.lane8); A.w[j] = st_ld(w.wb, (rj << 3) + w.lane8); }
        if (MCOL) A.cm[j] = w.cmask[mr + (uint32_t)j * w.mline + w.lane];
    }
}

// requests of the step at plane z: plane z + 2 of the own rows into pq, and the rest of plane z + 1 into A
// (past the last plane of the vector the requests repeat the last one: no tests, nothing of it is used)
template <int NDOT, int R, int EPI, bool MCOL, int ND>
__device__ __forceinline__ void st_request(const StWalk<R, ND> &w, unsigned z, double (&pq)[R], StAux<R, NDOT, EPI, MCOL> &A)
{
    const bool more = z + 1u < w.nz;
    const uint32_t r = more ? w.rz + w.sz : w.rz, mr = more ? w.mz + w.mstep : w.mz;
    const uint32_t rq = z + 2u < w.nz ? w.rz + 2u * w.sz : r;
#pragma unroll
    for (int j = 0; j < R; ++j) pq[j] = st_ld(w.xb, ((rq + (uint32_t)j * w.sy) << 3) + w.lane8);
    st_aux<NDOT, R, EPI, MCOL, ND>(w, r, mr, A);
}

// the R slices of plane z: own rows pc, the planes below / above pm / pp, the rest in cur; then on to the next plane
template <int NDOT, int R, int EPI, bool MCOL, int ND>
__device__ __forceinline__ void st_plane(StWalk<R, ND> &w, unsigned z, const double (&pm)[R], const double (&pc)[R], const double (&pp)[R],
                                         const StAux<R, NDOT, EPI, MCOL> &cur, const typename st_codes<R>::type &ccur)
{
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t cj = ccur[j];
        if (cj != w.cidx) {
            w.cidx = cj;
            const BICG_KCONST StencilTab *t = w.tab + cj;
#pragma unroll
            for (int k = 0; k < 7; ++k) w.cv[k] = t->v[k];
            w.cbits = (uint32_t)t->bits;
        }
        double xv[7];
        xv[0] = pm[j];
        xv[1] = j > 0 ? pc[j > 0 ? j - 1 : 0] : cur.lo;
        xv[2] = st_wave_shift<0x138>(pc[j], cur.e[j]);
        xv[3] = pc[j];
        xv[4] = st_wave_shift<0x130>(pc[j], cur.e[j]);
        xv[5] = j < R - 1 ? pc[j < R - 1 ? j + 1 : 0] : cur.hi;
        xv[6] = pp[j];
        double sum;
        if (MCOL) sum = st_sum_bits(w.cv, xv, cur.cm[j]);
        else if (w.cbits == 0x7Fu) sum = st_sum_all(w.cv, xv);
        else sum = st_sum_bits(w.cv, xv, w.cbits);
        const double yi = 0.0 + sum;                                  // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
        const uint32_t off = ((w.rz + (uint32_t)j * w.sy) << 3) + w.lane8;
        st_st(w.yb, off, yi, w.nt);
        if (EPI) {
            const double qv = cur.r[j] + (-w.alpha) * pc[j];          // q = r - alpha s            (src/solver.c:225-226)
            const double yv = cur.w[j] + (-w.alpha) * yi;             // y = w - alpha z            (src/solver.c:227-228)
            st_st(w.rb, off, qv, w.nt);
            st_st(w.wb, off, yv, w.nt);
            w.acc[0] += qv * yv;
            w.acc[ND > 1 ? 1 : 0] += yv * yv;
        } else if (NDOT >= 1) {
            const double ume = cur.u[j];
            w.acc[0] += ume * yi;
            if (NDOT == 2) w.acc[ND > 1 ? 1 : 0] += yi * yi;
            if (NDOT == 3) w.acc[ND > 1 ? 1 : 0] += ume * ume;
        }
    }
    if (z + 1u < w.nz) { w.rz += w.sz; w.mz += w.mstep; w.code += w.cstep; }
}

template <int NDOT, int R, int EPI, bool MCOL>
__device__ __forceinline__ void stencil_tile(const SpmvArgs &a, unsigned xs, unsigned y0, unsigned z0, unsigned z1, unsigned lane,
                                             double (&acc)[(NDOT > 0 || EPI) ? (EPI ? 2 : NDOT) : 1])
{
    constexpr int ND = (NDOT > 0 || EPI) ? (EPI ? 2 : NDOT) : 1;
    const StencilDev &g = a.sell.st;
    StWalk<R, ND> w;
    w.xb = reinterpret_cast<const char *>(a.x); w.yb = reinterpret_cast<char *>(a.y); w.ub = reinterpret_cast<const char *>(a.u);
    w.rb = reinterpret_cast<char *>(a.epi.r); w.wb = reinterpret_cast<char *>(a.epi.w);
    w.cmask = g.cmask; w.tab = (const BICG_KCONST StencilTab *)g.tab;
    w.n = a.nrows; w.sy = g.sy; w.sz = g.sz; w.nz = g.nz;
    w.lane = lane; w.lane8 = lane << 3; w.hi_half = lane < 32u ? 0u : 0xFFFFFFFFu;
    w.alpha = EPI ? a.S->alpha : 0.0;
    w.nt = g.nt_store;
    const unsigned dense = (unsigned)__builtin_popcountll(g.mcols & ((1ull << xs) - 1ull));
    // rows are scalars: rz = first row of the tile's line 0 in the current plane; line j is j sy rows further
    w.rz = (z0 * g.ny + y0) * g.sy + xs * (uint32_t)kSliceRows;
    w.mline = g.nmc * (uint32_t)kSliceRows;
    w.mz = ((z0 * g.ny + y0) * g.nmc + dense) * (uint32_t)kSliceRows;
    w.mstep = g.ny * w.mline;
    w.cstep = g.nxs * g.ny;
    w.code = (const BICG_KCONST unsigned *)g.code + ((size_t)(z0 * g.nxs + xs) * g.ny + y0);
    w.cidx = 0xFFFFFFFFu; w.cbits = 0u;
#pragma unroll
    for (int k = 0; k < 7; ++k) w.cv[k] = 0.0;
#pragma unroll
    for (int d = 0; d < ND; ++d) w.acc[d] = 0.0;
    typedef typename st_codes<R>::type codes_t;
    typedef StAux<R, NDOT, EPI, MCOL> aux_t;

    // Four sets of plane registers and two of everything else, the loop unrolled four times: no value is ever COPIED from the
    // register it was loaded into (a rotation by moves makes the end of every step a wait for the loads it has just issued)
    double p0[R], p1[R], p2[R], p3[R];
    aux_t xa, xb2;
    codes_t ca, cb;
    {
        // planes z0 - 1, z0, z0 + 1 and the rest of plane z0: what a step in front of the tile would have requested
        const uint32_t rm = z0 > 0u ? w.rz - w.sz : w.rz, rp = z0 + 1u < w.nz ? w.rz + w.sz : w.rz;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            p0[j] = st_ld(w.xb, ((rm + (uint32_t)j * w.sy) << 3) + w.lane8);
            p1[j] = st_ld(w.xb, ((w.rz + (uint32_t)j * w.sy) << 3) + w.lane8);
            p2[j] = st_ld(w.xb, ((rp + (uint32_t)j * w.sy) << 3) + w.lane8);
        }
        st_aux<NDOT, R, EPI, MCOL, ND>(w, w.rz, w.mz, xa);
        ca = *(const BICG_KCONST codes_t *)w.code;
    }
#define ST_STEP(PM, PC, PP, PQ, CUR, NXT, CC, CN)                                                           \
    {                                                                                                       \
        st_request<NDOT, R, EPI, MCOL, ND>(w, z, PQ, NXT);                                                  \
        CN = *(const BICG_KCONST codes_t *)(z + 1u < w.nz ? w.code + w.cstep : w.code);                     \
        /* the requests stay HERE, in front of this plane's arithmetic: left to itself the compiler sinks them behind the first */ \
        /* line's sum and waits for everything -- the previous plane's stores included -- before it */     \
        asm volatile("" ::: "memory");                                                                      \
        st_plane<NDOT, R, EPI, MCOL, ND>(w, z, PM, PC, PP, CUR, CC);                                        \
        ++z;                                                                                                \
    }
    for (unsigned z = z0; z < z1;) {
        ST_STEP(p0, p1, p2, p3, xa, xb2, ca, cb)
        if (z >= z1) break;
        ST_STEP(p1, p2, p3, p0, xb2, xa, cb, ca)
        if (z >= z1) break;
        ST_STEP(p2, p3, p0, p1, xa, xb2, ca, cb)
        if (z >= z1) break;
        ST_STEP(p3, p0, p1, p2, xb2, xa, cb, ca)
    }
#undef ST_STEP
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = w.acc[d];
}

template <int NDOT, int R, int MODE, int EPI>
__global__ void __launch_bounds__(kBlock) k_spmv_stencil(SpmvArgs a)
{
    constexpr int ND = EPI ? 2 : (NDOT > 0 ? NDOT : 1);
    constexpr bool DOTS = EPI || NDOT > 0;
    const int done = a.S->done;
    __shared__ double sm[5 * ND];
    const unsigned bid = blockIdx.x, nblocks = gridDim.x;
    if (MODE == RED_WAVE) {
        // a dot group of EARLIER kernels rides on this launch (see k_spmv_sell)
        __shared__ FinishLds fl;
        if (a.fin.seq && (bid < (unsigned)kShards || (a.fin.roles & FIN_APPLY))) (void)finish_group(a.S, a.fin, a.fin.roles, bid, nblocks, fl, nullptr);
    }
    double acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = 0.0;

    // Virtual workgroup vb of the canonical order: XCD-contiguous (workgroup b runs on XCD b % 8: one L2 serves the tiles that
    // share halo lines) and, every other product, reversed; its partial sums go to slot vb whichever physical workgroup it is.
    // vb -> (x segment, block of 4 R lines, block of zl planes), x segment fastest: neighbours in vb are neighbours in the grid.
    const StencilDev &g = a.sell.st;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const unsigned nyw = (g.ny + 4u * (unsigned)R - 1u) / (4u * (unsigned)R);
    unsigned vb = bid;
    if (g.xcd == 2) {
        // Sweep order (the plan sets it when nyw is a multiple of 8): XCD k owns the line blocks k nyw/8 .. (k+1) nyw/8 - 1 of every
        // plane and takes them plane block by plane block -- the tiles that share halo lines are dispatched together, to ONE L2, and
        // short (few planes per tile), so that the neighbour's rows are still there when a tile asks for them.
        const unsigned b = a.reverse ? nblocks - 1u - bid : bid, k = b % 8u, j = b / 8u, nywx = nyw / 8u;
        vb = ((j / (g.nxs * nywx)) * nyw + k * nywx + (j / g.nxs) % nywx) * g.nxs + j % g.nxs;
    } else {
        if (g.xcd && bid < (nblocks / 8u) * 8u) vb = (bid % 8u) * (nblocks / 8u) + bid / 8u;
        if (a.reverse) vb = nblocks - 1u - vb;
    }
    const unsigned xs = vb % g.nxs, yw = (vb / g.nxs) % nyw, zb = vb / (g.nxs * nyw);
    const unsigned y0 = (yw * 4u + wave) * (unsigned)R;
    const unsigned z0 = g.z_lo + zb * g.zl, z1 = z0 + g.zl < g.z_hi ? z0 + g.zl : g.z_hi;
    if (!done && y0 < g.ny) {              // (done: nothing is stored and nothing published)
        if ((g.mcols >> xs) & 1ull) stencil_tile<NDOT, R, EPI, true>(a, xs, y0, z0, z1, lane, acc);
        else stencil_tile<NDOT, R, EPI, false>(a, xs, y0, z0, z1, lane, acc);
    }
    if (DOTS && !done) {
        if (MODE == RED_WAVE) wave_publish<ND>(acc, a.red.partial, a.red.slot_base + vb);
        else reduce_publish<ND, MODE == RED_TICKET_HEAVY>(acc, a.S, a.red, a.red.slot_base + vb, sm, a.red.slot_base + bid);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The wide form (round 6): W = 2 or 4 consecutive rows per lane, a wavefront's line is 64 W rows (W x segments).
//
// What it is for: at 512^3 the product above is bound by what crosses the fabric (1.9 GB read + 1.07 GB written at 6.5 TB/s), and
// of its 8.6 L2 requests per slice 7.1 miss: the four own lines, and nearly every line fetched for ANOTHER workgroup's rows -- the
// two edge lines of every slice (128 bytes for 8) and the line below / above a workgroup's 16 lines. Neither the order of the tiles
// (an XCD sweeping its own range of lines plane block by plane block) nor fewer resident workgroups changed that
// (profiles/r06/stencil_notes.txt): neighbours drift apart by more planes than an L2 holds. So the tile itself gets wider: with W
// rows per lane a wavefront has ONE pair of edge lines per 64 W rows, the +-1 entries of the inner rows are the lane's own
// registers, and the accesses are 16 / 32 bytes per lane.
//
// Restriction (the plan checks it, StencilDev::wide): ONE value per canonical position in the whole block -- all table entries
// agree wherever they have an entry (constant-coefficient stencils: the Laplacian of BASELINE.json configs[3]) -- because the
// segments of a line may then differ in WHICH entries their rows have only: a byte of presence bits per segment
// (StencilDev::wbits, one word per wavefront line) or per row (cmask, in masked x segments) instead of a table index per slice.
// Arithmetic, order and rounding are those of the narrow form: bit for bit mult() (src/matrix.c:506-515).
template <int W> struct st_vec;
template <> struct st_vec<2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct st_vec<4> { typedef double type __attribute__((ext_vector_type(4))); };
template <int W> struct st_full;
template <> struct st_full<2> { static constexpr uint32_t v = 0x7F7Fu; };
template <> struct st_full<4> { static constexpr uint32_t v = 0x7F7F7F7Fu; };

template <int W> __device__ __forceinline__ typename st_vec<W>::type stw_ld(const char *base, uint32_t off)
{
    return *reinterpret_cast<const typename st_vec<W>::type *>(base + off);
}
template <int W> __device__ __forceinline__ void stw_st(char *base, uint32_t off, typename st_vec<W>::type v, int nt)
{
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<typename st_vec<W>::type *>(base + off));
    else *reinterpret_cast<typename st_vec<W>::type *>(base + off) = v;
}

template <int R, int W, int NDOT, int EPI, bool MCOL> struct StAuxW {
    typedef typename st_vec<W>::type vec;
    vec lo, hi;                          // x of the line below line 0 / above line R - 1
    double e[R];                         // lanes 0..31: x left of the wavefront's line, lanes 32..63: right of it
    vec u[(NDOT > 0 && !EPI) ? R : 1];
    vec r[EPI ? R : 1], w[EPI ? R : 1];
    uint32_t cm[MCOL ? R : 1];           // W bytes of row bits (masked x segments)
};

template <int R, int W, int ND> struct StWalkW {
    const char *xb, *ub; char *yb, *rb, *wb;
    const unsigned char *cmask;
    const BICG_KCONST unsigned *wbits;      // the R words of the current plane
    uint32_t n, sy, sz, nz, laneb, hi_half;
    uint32_t rz, mz, mline, mstep;          // first row of line 0 in the current plane; the same place in cmask
    uint32_t cstep;                         // words per plane
    uint32_t cm_off, cm_use, sub8;          // per lane: where its rows' bits lie in a cmask line / whether they do / 8 x its segment of the line
    int nt;
    double cv[7];
    double alpha;
    double acc[ND];
};

template <int NDOT, int R, int W, int EPI, bool MCOL, int ND>
__device__ __forceinline__ void stw_aux(const StWalkW<R, W, ND> &w, uint32_t r, uint32_t mr, StAuxW<R, W, NDOT, EPI, MCOL> &A)
{
    constexpr uint32_t kLine = (uint32_t)kSliceRows * W;
    const uint32_t rlo = r >= w.sy ? r - w.sy : r, rhi = r + (uint32_t)R * w.sy < w.n ? r + (uint32_t)R * w.sy : r + (uint32_t)(R - 1) * w.sy;
    A.lo = stw_ld<W>(w.xb, (rlo << 3) + w.laneb);
    A.hi = stw_ld<W>(w.xb, (rhi << 3) + w.laneb);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t rj = r + (uint32_t)j * w.sy;
        const uint32_t left = rj > 0u ? rj - 1u : rj, right = rj + kLine < w.n ? rj + kLine : rj + kLine - 1u;
        A.e[j] = st_ld(w.xb, (left << 3) + (w.hi_half & ((right - left) << 3)));
        if (NDOT > 0 && !EPI) A.u[j] = stw_ld<W>(w.ub, (rj << 3) + w.laneb);
        if (EPI) { A.r[j] = stw_ld<W>(w.rb, (rj << 3) + w.laneb); A.w[j] = stw_ld<W>(w.wb, (rj << 3) + w.laneb); }
        if (MCOL) {
            const unsigned char *at = w.cmask + (mr + (uint32_t)j * w.mline + w.cm_off);
            A.cm[j] = W == 2 ? (uint32_t)*reinterpret_cast<const unsigned short *>(at) : *reinterpret_cast<const uint32_t *>(at);
        }
    }
}

template <int NDOT, int R, int W, int EPI, bool MCOL, int ND>
__device__ __forceinline__ void stw_request(const StWalkW<R, W, ND> &w, unsigned z, typename st_vec<W>::type (&pq)[R], StAuxW<R, W, NDOT, EPI, MCOL> &A)
{
    const bool more = z + 1u < w.nz;
    const uint32_t r = more ? w.rz + w.sz : w.rz, mr = more ? w.mz + w.mstep : w.mz;
    const uint32_t rq = z + 2u < w.nz ? w.rz + 2u * w.sz : r;
#pragma unroll
    for (int j = 0; j < R; ++j) pq[j] = stw_ld<W>(w.xb, ((rq + (uint32_t)j * w.sy) << 3) + w.laneb);
    stw_aux<NDOT, R, W, EPI, MCOL, ND>(w, r, mr, A);
}

template <int NDOT, int R, int W, int EPI, bool MCOL, int ND>
__device__ __forceinline__ void stw_plane(StWalkW<R, W, ND> &w, unsigned z, const typename st_vec<W>::type (&pm)[R], const typename st_vec<W>::type (&pc)[R],
                                          const typename st_vec<W>::type (&pp)[R], const StAuxW<R, W, NDOT, EPI, MCOL> &cur, const typename st_codes<R>::type &wcur)
{
    typedef typename st_vec<W>::type vec;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t wb = wcur[j];
        const bool all = !MCOL && wb == st_full<W>::v;                  // (a scalar: every row of the line has all seven entries)
        uint32_t lb = ((wb >> w.sub8) & 0x7Fu) * 0x01010101u;            // the lane's segment's bits, once per row of the lane
        if (MCOL) lb = w.cm_use ? cur.cm[j] : lb;
        const double left = st_wave_shift<0x138>(pc[j][W - 1], cur.e[j]), right = st_wave_shift<0x130>(pc[j][0], cur.e[j]);
        vec yv, qv, tv;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            double xv[7];
            xv[0] = pm[j][k];
            xv[1] = j > 0 ? pc[j > 0 ? j - 1 : 0][k] : cur.lo[k];
            xv[2] = k > 0 ? pc[j][k > 0 ? k - 1 : 0] : left;
            xv[3] = pc[j][k];
            xv[4] = k < W - 1 ? pc[j][k < W - 1 ? k + 1 : 0] : right;
            xv[5] = j < R - 1 ? pc[j < R - 1 ? j + 1 : 0][k] : cur.hi[k];
            xv[6] = pp[j][k];
            const double sum = all ? st_sum_all(w.cv, xv) : st_sum_bits(w.cv, xv, (lb >> (8 * k)) & 0x7Fu);
            const double yi = 0.0 + sum;                                  // y = 0 ; y += tempy  (src/matrix.c:434-437, 514)
            yv[k] = yi;
            if (EPI) {
                const double q = cur.r[j][k] + (-w.alpha) * pc[j][k];     // q = r - alpha s            (src/solver.c:225-226)
                const double t = cur.w[j][k] + (-w.alpha) * yi;           // y = w - alpha z            (src/solver.c:227-228)
                qv[k] = q; tv[k] = t;
                w.acc[0] += q * t;
                w.acc[ND > 1 ? 1 : 0] += t * t;
            } else if (NDOT >= 1) {
                const double ume = cur.u[j][k];
                w.acc[0] += ume * yi;
                if (NDOT == 2) w.acc[ND > 1 ? 1 : 0] += yi * yi;
                if (NDOT == 3) w.acc[ND > 1 ? 1 : 0] += ume * ume;
            }
        }
        const uint32_t off = ((w.rz + (uint32_t)j * w.sy) << 3) + w.laneb;
        stw_st<W>(w.yb, off, yv, w.nt);
        if (EPI) { stw_st<W>(w.rb, off, qv, w.nt); stw_st<W>(w.wb, off, tv, w.nt); }
    }
    if (z + 1u < w.nz) { w.rz += w.sz; w.mz += w.mstep; w.wbits += w.cstep; }
}

template <int NDOT, int R, int W, int EPI, bool MCOL>
__device__ __forceinline__ void stencil_tile_w(const SpmvArgs &a, unsigned xs, unsigned y0, unsigned z0, unsigned z1, unsigned lane,
                                               double (&acc)[(NDOT > 0 || EPI) ? (EPI ? 2 : NDOT) : 1])
{
    constexpr int ND = (NDOT > 0 || EPI) ? (EPI ? 2 : NDOT) : 1;
    typedef typename st_vec<W>::type vec;
    const StencilDev &g = a.sell.st;
    const uint32_t nxw = g.nxs / (uint32_t)W;
    StWalkW<R, W, ND> w;
    w.xb = reinterpret_cast<const char *>(a.x); w.yb = reinterpret_cast<char *>(a.y); w.ub = reinterpret_cast<const char *>(a.u);
    w.rb = reinterpret_cast<char *>(a.epi.r); w.wb = reinterpret_cast<char *>(a.epi.w);
    w.cmask = g.cmask;
    w.n = a.nrows; w.sy = g.sy; w.sz = g.sz; w.nz = g.nz;
    w.laneb = lane * (8u * W); w.hi_half = lane < 32u ? 0u : 0xFFFFFFFFu;
    w.alpha = EPI ? a.S->alpha : 0.0;
    w.nt = g.nt_store;
    // the lane's W rows lie in segment seg of the line's W segments, from row r0 of it on
    const uint32_t sub = (lane * (uint32_t)W) >> 6, r0 = (lane * (uint32_t)W) & 63u, seg = xs * (uint32_t)W + sub;
    w.sub8 = 8u * sub;
    w.cm_use = (uint32_t)((g.mcols >> seg) & 1ull);
    w.cm_off = (w.cm_use ? (uint32_t)__builtin_popcountll(g.mcols & ((1ull << seg) - 1ull)) * (uint32_t)kSliceRows : 0u) + r0;
    w.rz = (z0 * g.ny + y0) * g.sy + xs * (uint32_t)(kSliceRows * W);
    w.mline = g.nmc * (uint32_t)kSliceRows;
    w.mz = (z0 * g.ny + y0) * w.mline;
    w.mstep = g.ny * w.mline;
    w.cstep = nxw * g.ny;
    w.wbits = (const BICG_KCONST unsigned *)g.wbits + ((size_t)(z0 * nxw + xs) * g.ny + y0);
    {
        const BICG_KCONST StencilTab *t = (const BICG_KCONST StencilTab *)g.tab + g.ref;
#pragma unroll
        for (int k = 0; k < 7; ++k) w.cv[k] = t->v[k];
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) w.acc[d] = 0.0;
    typedef typename st_codes<R>::type codes_t;
    typedef StAuxW<R, W, NDOT, EPI, MCOL> aux_t;

    vec p0[R], p1[R], p2[R], p3[R];         // (four sets of plane registers, renamed instead of moved: see stencil_tile)
    aux_t xa, xb2;
    codes_t ca, cb;
    {
        const uint32_t rm = z0 > 0u ? w.rz - w.sz : w.rz, rp = z0 + 1u < w.nz ? w.rz + w.sz : w.rz;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            p0[j] = stw_ld<W>(w.xb, ((rm + (uint32_t)j * w.sy) << 3) + w.laneb);
            p1[j] = stw_ld<W>(w.xb, ((w.rz + (uint32_t)j * w.sy) << 3) + w.laneb);
            p2[j] = stw_ld<W>(w.xb, ((rp + (uint32_t)j * w.sy) << 3) + w.laneb);
        }
        stw_aux<NDOT, R, W, EPI, MCOL, ND>(w, w.rz, w.mz, xa);
        ca = *(const BICG_KCONST codes_t *)w.wbits;
    }
#define STW_STEP(PM, PC, PP, PQ, CUR, NXT, CC, CN)                                                          \
    {                                                                                                       \
        stw_request<NDOT, R, W, EPI, MCOL, ND>(w, z, PQ, NXT);                                              \
        CN = *(const BICG_KCONST codes_t *)(z + 1u < w.nz ? w.wbits + w.cstep : w.wbits);                   \
        asm volatile("" ::: "memory");                                                                      \
        stw_plane<NDOT, R, W, EPI, MCOL, ND>(w, z, PM, PC, PP, CUR, CC);                                    \
        ++z;                                                                                                \
    }
    for (unsigned z = z0; z < z1;) {
        STW_STEP(p0, p1, p2, p3, xa, xb2, ca, cb)
        if (z >= z1) break;
        STW_STEP(p1, p2, p3, p0, xb2, xa, cb, ca)
        if (z >= z1) break;
        STW_STEP(p2, p3, p0, p1, xa, xb2, ca, cb)
        if (z >= z1) break;
        STW_STEP(p3, p0, p1, p2, xb2, xa, cb, ca)
    }
#undef STW_STEP
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = w.acc[d];
}

template <int NDOT, int R, int W, int MODE, int EPI>
__global__ void __launch_bounds__(kBlock) k_spmv_stencil_w(SpmvArgs a)
{
    constexpr int ND = EPI ? 2 : (NDOT > 0 ? NDOT : 1);
    constexpr bool DOTS = EPI || NDOT > 0;
    const int done = a.S->done;
    __shared__ double sm[5 * ND];
    const unsigned bid = blockIdx.x, nblocks = gridDim.x;
    if (MODE == RED_WAVE) {
        __shared__ FinishLds fl;
        if (a.fin.seq && (bid < (unsigned)kShards || (a.fin.roles & FIN_APPLY))) (void)finish_group(a.S, a.fin, a.fin.roles, bid, nblocks, fl, nullptr);
    }
    double acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = 0.0;
    const StencilDev &g = a.sell.st;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const unsigned nxw = g.nxs / (unsigned)W, nyw = (g.ny + 4u * (unsigned)R - 1u) / (4u * (unsigned)R);
    unsigned vb = bid;                      // (the orders of k_spmv_stencil)
    if (g.xcd == 2) {
        const unsigned b = a.reverse ? nblocks - 1u - bid : bid, k = b % 8u, j = b / 8u, nywx = nyw / 8u;
        vb = ((j / (nxw * nywx)) * nyw + k * nywx + (j / nxw) % nywx) * nxw + j % nxw;
    } else {
        if (g.xcd && bid < (nblocks / 8u) * 8u) vb = (bid % 8u) * (nblocks / 8u) + bid / 8u;
        if (a.reverse) vb = nblocks - 1u - vb;
    }
    const unsigned xs = vb % nxw, yw = (vb / nxw) % nyw, zb = vb / (nxw * nyw);
    const unsigned y0 = (yw * 4u + wave) * (unsigned)R;
    const unsigned z0 = g.z_lo + zb * g.zl, z1 = z0 + g.zl < g.z_hi ? z0 + g.zl : g.z_hi;
    if (!done && y0 < g.ny) {
        if ((g.mcols >> (xs * (unsigned)W)) & ((1ull << W) - 1ull)) stencil_tile_w<NDOT, R, W, EPI, true>(a, xs, y0, z0, z1, lane, acc);
        else stencil_tile_w<NDOT, R, W, EPI, false>(a, xs, y0, z0, z1, lane, acc);
    }
    if (DOTS && !done) {
        if (MODE == RED_WAVE) wave_publish<ND>(acc, a.red.partial, a.red.slot_base + vb);
        else reduce_publish<ND, MODE == RED_TICKET_HEAVY>(acc, a.S, a.red, a.red.slot_base + vb, sm, a.red.slot_base + bid);
    }
}

unsigned stencil_grid(const StencilDev &st)
{
    if (!st.on) return 0u;
    const unsigned nyw = (st.ny + 4u * st.lines - 1u) / (4u * st.lines), nzb = (st.z_hi - st.z_lo + st.zl - 1u) / st.zl;
    return (st.wide ? st.nxs / st.wide : st.nxs) * nyw * nzb;
}

template <class K>
static void stencil_go(K kernel, const SpmvArgs &a, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    const dim3 g(stencil_grid(a.sell.st)), b(kBlock);
    if (e0 && e1) hipExtLaunchKernelGGL(kernel, g, b, 0, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kernel, g, b, 0, st, a);
    static const bool debug = getenv("BICG_DEBUG") != nullptr;
    if (debug) {
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) fprintf(stderr, "bicgstab_hip: HIP error \"%s\" noticed at: k_spmv_stencil\n", hipGetErrorString(err));
    }
}

template <int R>
static bool stencil_launch_lines(const SpmvArgs &a, int ndot, int epi, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    const int mode = red_mode(a.red, a.fin, ndot > 0 || epi);
#define ST_MODE(ND, EP)                                                                                         \
    do {                                                                                                        \
        if (mode == RED_WAVE) stencil_go(k_spmv_stencil<ND, R, RED_WAVE, EP>, a, st, e0, e1);                   \
        else if (mode == RED_TICKET_HEAVY) stencil_go(k_spmv_stencil<ND, R, ((ND) > 0 || (EP)) ? RED_TICKET_HEAVY : RED_TICKET, EP>, a, st, e0, e1); \
        else stencil_go(k_spmv_stencil<ND, R, RED_TICKET, EP>, a, st, e0, e1);                                  \
    } while (0)
    if (epi) {
        if (mode == RED_WAVE) return false;                       // (CA-BiCGStab applies its scalars in the producer: ticket modes only)
        ST_MODE(0, 1);
        return true;
    }
    if (ndot == 0) ST_MODE(0, 0); else if (ndot == 1) ST_MODE(1, 0); else if (ndot == 2) ST_MODE(2, 0); else ST_MODE(3, 0);
#undef ST_MODE
    return true;
}

template <int R, int W>
static bool stencil_launch_wide(const SpmvArgs &a, int ndot, int epi, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    const int mode = red_mode(a.red, a.fin, ndot > 0 || epi);
#define STW_MODE(ND, EP)                                                                                        \
    do {                                                                                                        \
        if (mode == RED_WAVE) stencil_go(k_spmv_stencil_w<ND, R, W, RED_WAVE, EP>, a, st, e0, e1);              \
        else if (mode == RED_TICKET_HEAVY) stencil_go(k_spmv_stencil_w<ND, R, W, ((ND) > 0 || (EP)) ? RED_TICKET_HEAVY : RED_TICKET, EP>, a, st, e0, e1); \
        else stencil_go(k_spmv_stencil_w<ND, R, W, RED_TICKET, EP>, a, st, e0, e1);                             \
    } while (0)
    if (epi) {
        if (mode == RED_WAVE) return false;
        STW_MODE(0, 1);
        return true;
    }
    if (ndot == 0) STW_MODE(0, 0); else if (ndot == 1) STW_MODE(1, 0); else if (ndot == 2) STW_MODE(2, 0); else STW_MODE(3, 0);
#undef STW_MODE
    return true;
}

// the wide form reads and writes 8 W bytes per lane: every vector it touches must be aligned to that
static bool stencil_wide_ok(const SpmvArgs &a, int ndot, int epi)
{
    const uintptr_t m = 8u * a.sell.st.wide - 1u;
    uintptr_t bits = reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y);
    if (ndot > 0 && !epi) bits |= reinterpret_cast<uintptr_t>(a.u);
    if (epi) bits |= reinterpret_cast<uintptr_t>(a.epi.r) | reinterpret_cast<uintptr_t>(a.epi.w);
    return (bits & m) == 0;
}

bool launch_spmv_stencil(const SpmvArgs &a, int ndot, int epi, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    if (!a.sell.st.on || a.has_shift) return false;      // (shifted products: the slice-by-slice loop)
    if (a.sell.st.wide) {
        if (!stencil_wide_ok(a, ndot, epi)) {
            fprintf(stderr, "ERROR: bicgstab_hip: the wide plane-marching product was planned, but a vector of this launch is not aligned to %u bytes\n", 8u * a.sell.st.wide);
            abort();
        }
        const unsigned key = a.sell.st.lines * 10u + a.sell.st.wide;
        if (key == 22u || key == 24u || key == 42u || key == 44u) g_product_kernels |= PK_STENCIL;
        switch (key) {
        case 22: return stencil_launch_wide<2, 2>(a, ndot, epi, st, e0, e1);
        case 24: return stencil_launch_wide<2, 4>(a, ndot, epi, st, e0, e1);
        case 42: return stencil_launch_wide<4, 2>(a, ndot, epi, st, e0, e1);
        case 44: return stencil_launch_wide<4, 4>(a, ndot, epi, st, e0, e1);
        default: return false;
        }
    }
    if (a.sell.st.lines == 2 || a.sell.st.lines == 4) g_product_kernels |= PK_STENCIL;
    switch (a.sell.st.lines) {
    case 2: return stencil_launch_lines<2>(a, ndot, epi, st, e0, e1);
    case 4: return stencil_launch_lines<4>(a, ndot, epi, st, e0, e1);
    default: return false;
    }
}

void preload_stencil_kernels()
{
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k_spmv_stencil<0, 4, RED_TICKET, 0>));
    (void)hipGetLastError();
}

}  // namespace bicg
