// bicg_device.h -- device-side state and kernel launch interface shared by the kernels
// (bicg_kernels.hip) and the host side (bicg_solver.cpp, bicg_shifted.cpp, bicg_create.cpp: bicg_host.h). gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bicg {

constexpr int kBlock = 256;                 // 4 wavefronts of 64
constexpr int kNnzPerThread = 8;            // SpMV: products staged per thread
constexpr int kChunk = kBlock * kNnzPerThread;  // 2048 products (16 KiB of LDS) staged per row block
constexpr int kRowBlockNnz = kChunk - 4;        // row blocks hold at most this many non-zeros: the staged
                                                // window starts at a multiple of 4 entries (16-byte loads)
constexpr int kPadEntries = 4;                  // val/col device arrays are padded by this many entries
constexpr int kMaxDots = 5;                 // widest dot group (pipelined phase 2)
constexpr int kRedSlots = 8;                // packed all-reduce buffer, doubles
constexpr int kPartialStride = 8;           // doubles per block in the partial-sum table (64 B)
constexpr int kShardLL = 33;                // rows of the LL shard-total table: kShards totals + one row of applied scalars
constexpr int kShards = 32;                 // arrival counters per dot group (one word saturates at ~88 atomics/us)
constexpr int kCounterStride = 32;          // unsigneds between shard counters (128 B apart)
constexpr int kMaxGrid = 2048;              // element-wise kernels: 256 CUs x 8 resident workgroups, grid-stride beyond
constexpr int kSpmvMaxGrid = 1 << 18;       // SpMV: one workgroup per row block up to this many
constexpr int kMailRing = 8;                // peer-to-peer all-reduce mailboxes: groups in flight before a slot is reused
constexpr int kHaloRing = 8;                // peer-to-peer halo landing zones: exchanges in flight before reuse
constexpr int kMaxRanksP2p = 64;            // ranks the peer-to-peer transport supports (one node)

// What the single thread that completes a dot group does with the (globally reduced) sums in
// Scal::red. One value per blocking point of the reference's loops.
enum Phase : int {
    PH_NONE = 0,
    PH_INIT,        // red[0]=(r,r): rTr = dot_r = dot_zero           (src/solver.c:78-83, 203, 336)
    PH_INIT_ALPHA,  // red[0]=(r,w): alpha = rTr/(r,w), beta=omega=0   (src/solver.c:210-211, 345-346)
    PH_PLAIN_ALPHA, // red[0]=(r#,s): alpha = rTr/(r#,s)               (src/solver.c:93)
    PH_OMEGA,       // red[0]=(q,y), red[1]=(y,y): omega               (src/solver.c:104, 232, 369)
    PH_PLAIN_END,   // red[0]=(r,r), red[1]=(r#,r): beta, k++          (src/solver.c:108-120)
    PH_RECUR_END,   // red[0..4]=(r,r),(r#,r),(r#,w),(r#,s),(r#,z): beta, alpha, k++ (src/solver.c:248-251, 387-390)
    // shifted BiCGStab (reference src/shifted_solver.c:182-354): applied by a whole workgroup,
    // one thread per shift for the per-shift scalar recurrences
    PH_SH_INIT,     // red[0]=(r,r): rTr = dot_r = dot_zero, per-shift scalars reset            (:238-255)
    PH_SH_ALPHA,    // red[0]=(r#,s): alpha[seed]; beta[j], pi, eta, alpha[j]                    (:264-287)
    PH_SH_OMEGA,    // red[0]=(q,y), red[1]=(q,q): omega[seed]; omega[j], x/p coefficients, zeta (:291-301)
    PH_SH_END,      // red[0]=(r,r), red[1]=(r#,r): beta[seed], max|1/(zeta pi)|, k++            (:304-320)
    // pipelined shifted variant (src/shifted_solver.c:703-895)
    PH_SHP_INIT_ALPHA, // red[0]=(r,w): alpha[seed] = rTr/(r,w), alpha_old = 1                   (:785-786)
    PH_SHP_OMEGA,      // red[0]=(q,y), red[1]=(y,y): omega[seed] and ALL per-shift scalars      (:803-839)
    PH_SHP_END,        // red[0..4]=(r,r),(r#,r),(r#,w),(r#,s),(r#,z): beta, alpha, max, k++     (:856-866)
    // shifted solvers with per-shift stop flags and seed switching
    // (reference src/shifted_switching_solver.c:20-257 and :260-608)
    PH_SW_INIT,     // red[0]=(r,r): rTr = dot_r = dot_zero, archives and per-shift state reset       (:344-364)
    PH_SW_ALPHA,    // red[0]=(r#,s): alpha archive[k]                                               (:389-392)
    PH_SW_OMEGA,    // red[0]=(q,y), red[1]=(q,q): omega archive[k]                                  (:407-412)
    PH_SW_END,      // red[0]=(r,r), red[1]=(r#,r): beta archive[k]; eta, pi, alpha_j, omega_j, the
                    // update coefficients, zeta, beta_j of every active shift                       (:416-446)
    PH_SW_STOP,     // no sums: stop flags, largest |1/(zeta pi)|, k++, seed switching (pauses)      (:451-536)
    PH_DRIFT,       // persistent kernels: red[0]=||(b - A x) - r||^2, red[1]=||r||^2: replacement iteration next?
};

enum ShiftMode : int {
    SH_LOP = 0,      // shifted_lopbicgstab (+_v2, _nooverlap)       src/shifted_solver.c:182-701
    SH_PIPE = 1,     // shifted_pipe_lopbicgstab (+_nooverlap)       src/shifted_solver.c:703-1086
    SH_XI = 2,       // shifted_bicgstab, seed 0, xi/tau recurrences src/shifted_solver.c:13-180
    SH_FLAG = 3,     // shifted_lopbicg: per-shift stop flags        src/shifted_switching_solver.c:20-257
    SH_SWITCH = 4,   // shifted_lopbicg_switching (+_noovlp)          src/shifted_switching_solver.c:260-1016
};

// Per-shift scalar state of the shifted solver, device resident (arrays of nsig doubles).
struct ShiftDev {
    int     nsig, seed, mode, pad;
    double  alpha_old, beta_old, max_zeta_pi;
    // SH_XI reuses the arrays: pi_old = xi_old, pi_new = xi_curr, eta = xi_new, zeta = tau
    double *sigma, *alpha, *beta, *omega, *eta, *zeta, *pi_old, *pi_new;
    // coefficients the batched update kernel reads for shift j (valid for one iteration)
    double *cp;   // 1 / (pi_new zeta)              p_j <- beta_j p_j + cp_j r_old      (:265-266)
    double *cx;   // omega_j / (pi_new zeta)        x_j += cx_j q                        (:296)
    double *c1;   // omega_j / (alpha_j zeta pi_new)        p_j += c1_j q                (:298)
    double *c2;   // -omega_j / (alpha_j zeta pi_old)       p_j += c2_j r_old            (:299)
    // SH_FLAG / SH_SWITCH (src/shifted_switching_solver.c): history of the seed's alpha/beta/omega
    // and of every shift's pi (index 0 = initial values, iteration k at index k), stop flags
    double *a_arc, *b_arc, *w_arc;   // [arc_len]
    double *pi_arc;                  // [nsig][arc_len]
    int    *stop;                    // [nsig] converged shifts (frozen)
    int    *skip;                    // [nsig] shifts the batched update of THIS iteration leaves alone
    int    *unsolved_arc;            // [arc_len] systems still running after iteration k (the reference's DISPLAY_SECTION_TIME column)
    int     arc_len, stop_count, max_sigma, switches;
    double  r_scale;                 // seed switch: r <- r_scale * r, applied by the host while paused  (:499)
};

// Device-resident scalar state of one solve. Kernels read alpha/beta/omega/done from here, so the
// host never has to synchronise inside an iteration.
struct Scal {
    double alpha, beta, omega;
    double rTr, rTr_old;
    double dot_r, dot_zero;
    double tol2;                 // tol*tol (src/solver.c:86)
    double red[kRedSlots];       // dot sums of the current group (local, then all-reduced in place)
    int    k;                    // iterations completed
    int    max_iter;
    int    done;                 // sticky: set when the reference's while condition fails
    int    breakdown_k;          // first iteration whose recurrence scalars were not finite (0 = none)
    double *tr_alpha, *tr_omega, *tr_beta, *tr_dotr;   // optional trace, [max_iter]
    ShiftDev *sh;                // shifted solver only
    int    comm_error;           // peer-to-peer transport: a wait for a peer timed out (sets done as well)
    int    paused;               // seed switching: done was raised only to hand control to the host (new seed pointers)
};
// A persistent pipelined launch (bicg_persist.hip) reports what it consumed and decided in the three slots of Scal::red no dot
// group uses (the block itself must not grow: every consumer-side kernel holds a private copy of it):
constexpr int kRedUsedV = 5;     // hand-offs (image tags, halo exchange numbers)
constexpr int kRedUsedG = 6;     // dot groups (table tags, mailbox numbers)
constexpr int kRedAdaptive = 7;  // replacement iterations the in-kernel drift check asked for

// Direct peer-to-peer transport (bicg_p2p.cpp). Values travel as "LL" words -- 8 bytes holding 32
// payload bits and the 32-bit sequence number of the operation, written with ONE store into
// memory of the receiving GPU (mapped through HIP IPC, uncached): a reader that sees the expected
// sequence number in a word has its payload too, so neither fences nor separate flags are needed.
// A double takes two words.
typedef unsigned long long llword;

// All-reduce of a dot group: the workgroup that completes the group's local sums stores them into
// the mailbox of EVERY rank (its own included); the apply kernel of the group waits for the P
// contributions and adds them in a fixed order, so all ranks obtain bit-identical sums.
// Mailbox layout: [kMailRing][nranks (source)][kRedSlots][2] words.
struct P2pRed {
    llword *const *mail;   // [nranks] every rank's mailbox as mapped in this process (device array)
    unsigned seq;          // sequence number of the group; 0 = no peer-to-peer publication
    unsigned mask;         // bit d set: this kernel publishes its sum d
    int rank, nranks;
    int n_collect;         // > 0 (with Reduce::apply_now): the finishing workgroup also waits for the n_collect
                           // sums of all ranks and applies the phase itself -- no separate apply kernel
    unsigned long long timeout_ticks;
};
__host__ __device__ inline size_t mail_index(unsigned seq, int nranks, int src, int d)
{
    return ((((size_t)(seq % kMailRing) * nranks + src) * kRedSlots) + d) * 2;
}

// Where a kernel's dot partial sums go and what happens when the last block has arrived.
struct Reduce {
    double   *partial;     // [slots][kPartialStride]
    double   *shard_tot;   // [kShards][kPartialStride]
    unsigned *counter;     // [(kShards + 1) * kCounterStride] arrival tickets, return to 0 after each group
    unsigned  expected;    // workgroups contributing to this group (possibly over two launches)
    unsigned  slot_base;   // slot of this launch's workgroup 0
    int       red_off;     // sums land in Scal::red[red_off + d]
    int       phase;       // Phase applied by the finishing thread when apply_now
    int       apply_now;   // single rank: apply the phase in-kernel; multi rank: host all-reduces first
    int       wave;        // 1: consumer-side finish -- store one partial per wavefront at row
                           // (slot_base + workgroup) * 4 + wave and end (see Finish); nothing else is used
    P2pRed    p2p;         // peer-to-peer transport: where the finished sums are published
    // Tail finish (single GPU, round 3): every workgroup stores its partials as LL words (tag tail_seq) and ENDS -- no
    // acknowledged store, no returning atomic (2-3 us of every workgroup's life on a Transport-sized product); the LAST
    // min(expected, kShards) workgroups of the group stay, add one shard of the table each (slot order) and publish the shard
    // totals as LL words, the very last one adds those and applies the recurrence. 0: arrival tickets as before.
    llword   *tail_tab;    // [slots][kTailStride] LL words
    llword   *tail_shard;  // [kShards][kRedSlots][2]
    unsigned  tail_seq;
};
constexpr int kTailStride = 16;             // LL words per slot of the tail table (8 doubles)

// ---- consumer-side finish of a dot group (the four solvers of reference src/solver.c) -------------
// A kernel that PRODUCES dot sums only stores one partial per wavefront (wave shuffle, one plain
// store, no barrier, no atomic, no ticket) and ends. The sums are completed by the kernel that
// CONSUMES the scalars: its first kShards workgroups add up one shard of the partials each and
// publish the shard totals as LL words (payload + sequence tag in one 8-byte store); every
// workgroup then waits for the kShards totals, adds them in shard order, applies the scalar
// recurrence (alpha = rTr/rTs ...) on a PRIVATE copy of the scalar block, and workgroup 0 writes
// that copy to the other of two alternating scalar blocks (late workgroups of the same launch
// still read the old one). The dependent chain partial -> ticket -> shard sum -> ticket -> total ->
// apply that used to end every producer (9 us of a 56 us Transport SpMV) is gone from the
// producers, and in the consumers it runs underneath their first vector loads.
// Summation order is fixed by slot numbers: bit-reproducible, whoever computes a shard.
enum FinishRole : int {
    FIN_SHARDS  = 1,   // workgroups < kShards sum the shards and publish the shard totals
    FIN_PUSH    = 2,   // peer-to-peer: workgroup 0 stores the local sums into every rank's mailbox
    FIN_APPLY   = 4,   // every workgroup: global sums -> recurrence on a private copy; workgroup 0 writes Snext
    FIN_BLOCK0  = 8,   // workgroup 0 only, IN PLACE on S: SpMV launches (phase PH_NONE: the other workgroups
                       // read nothing but `done`) and the stand-alone finisher (any phase)
    FIN_LOCAL   = 16,  // with FIN_BLOCK0: deposit this rank's sums only (the host enqueues an all-reduce)
};
struct Finish {
    const double *partial;   // [nparts][kPartialStride] one row per producing wavefront
    llword  *shard;          // [kShards][kRedSlots][2] LL words, tag = seq
    llword  *shard_clear;    // the previous group's LL words: zeroed by workgroup 0, so that a launch that is
                             // replayed with the same tag (hipGraph) never meets its own earlier words
    unsigned nparts;
    unsigned seq;            // 0 = nothing to finish
    int      n, red_off;     // sums land in red[red_off .. red_off + n)
    int      phase;          // Phase applied with FIN_APPLY
    int      roles;
    Scal    *Snext;          // FIN_APPLY: the scalar block later kernels read
    int     *alarm;          // peer-to-peer: raised when a wait for a peer timed out (every later wait returns at once)
    unsigned long long spin_ticks;   // wait this long (100 MHz) for a shard before summing it ourselves
    P2pRed   p2p;            // p2p.seq != 0: sums are exchanged through the mailboxes
};

// what every launch wrapper needs: the scalar block to read, the group to finish (if any), the
// stream, and which reduction epilogue the kernel is built with
enum RedMode : int { RED_TICKET = 0, RED_TICKET_HEAVY = 1, RED_WAVE = 2 };
struct Launch {
    Scal *S;
    Finish fin;
    hipStream_t st;
};

struct CsrDev {
    const double   *val;
    const uint32_t *col;
    const uint32_t *ptr;
};

// Sliced-ELL copy of the diag block: slices of 64 consecutive rows (one wavefront), entries stored
// slice by slice COLUMN-major -- entry k of row r sits at slice_base + k*64 + (r % 64) -- padded to
// the longest row of the slice with (val 0, col 0). Lane = row: val/col loads AND the x gather of
// a banded matrix are coalesced across the wavefront, each lane adds ITS row in stored order.
constexpr int kSliceRows = 64;
constexpr int kGroupRows = kBlock;          // one workgroup = 4 slices = 256 consecutive rows

// Plane-marching product (bicg_stencil.hip) for blocks whose slices are ALL list-driven (SellDev::all_lists) and whose lists are
// sub-sequences of ONE seven-entry list of distances (-sz, -sy, -1, 0, +1, +sy, +sz) in that (= stored, ascending column) order,
// with sy a multiple of 64 rows, sz a multiple of sy and the rows a multiple of sz: the 7-point stencil of an nx x ny x nz grid
// with nx = sy, whatever its coefficients and faces look like (BASELINE.json configs[3]). Nothing about the GRID is assumed
// beyond that: every x value is read at its literal distance from the row; what the lists do not contain is not added.
// A wavefront owns lines_per_wave consecutive grid lines of one 64-wide x segment and marches through `zl` planes: the values
// of its own rows in the planes z - 1, z, z + 1 stay in registers (the +-sz neighbours), the +-sy neighbours are the registers
// of the adjacent line or one halo load per side, the +-1 neighbours come from the adjacent lane (DPP wave shift) -- each x line
// is filled into an L1 once per wavefront that owns it plus (2 halo lines + 2 edge lines) per wavefront step, 6-8 cache lines per
// 64-row slice where the slice-by-slice product fills 20 (profiles/r04/laplace512_spmv_counters_lists_loop.txt).
//   code[(z nxs + xs) ny + y]   index of the slice's (distance list, value list) pair in tab
//   tab[i]                      {the seven values in canonical position (0.0 where the list has no entry), presence bits}
//   cmask                       one byte per row (presence bits) for the x segments that have masked slices (bit xs of mcols),
//                               at ((z ny + y) nmc + rank of xs among them) 64 + lane
struct StencilTab { double v[7]; unsigned long long bits; };      // 64 bytes: one scalar load
struct StencilDev {
    int on;
    uint32_t sy, sz;                // distances in rows
    uint32_t nxs, ny, nz;           // x segments (sy / 64), lines per plane (sz / sy), planes (rows / sz)
    uint32_t z_lo, z_hi;            // the planes this product takes (one rank: all; across ranks: the planes without halo entries)
    uint32_t zl;                    // planes per wavefront tile
    uint32_t lines;                 // lines per wavefront (2 or 4: the instantiations of k_spmv_stencil)
    uint32_t nmc;                   // x segments with masked slices
    int xcd;                        // XCD-contiguous order of the tiles (measurement knob BICG_STENCIL_XCD)
    int nt_store;                   // y (and the epilogue's vectors) stored non-temporally (measurement knob BICG_STENCIL_NT)
    unsigned long long mcols;
    const uint32_t *code;
    const StencilTab *tab;
    const unsigned char *cmask;
    // the wide form (k_spmv_stencil_w): rows per lane (0: the narrow form, 2 or 4), the table entry with all seven values, and one
    // word per wavefront line ((z nxs/wide + segment group) ny + y) with the presence bits of its `wide` segments, a byte each
    uint32_t wide, ref;
    const uint32_t *wbits;
};
struct SellDev {
    const double   *val;
    const uint32_t *col;
    const uint32_t *slice_base;   // [nslices] first entry of the slice
    const uint32_t *slice_len;    // [nslices] padded row length of the slice
    // optional 16-bit column offsets (col - row), slice by slice in quads: the four offsets of
    // entries 4q..4q+3 of a lane are contiguous -> element (q*64 + lane)*4 + (k % 4) from
    // slice_base16; null when some |col - row| >= 32768
    const short    *col16;
    const uint32_t *slice_base16;
    // jagged slices (ragged rows): step k of a slice stores the entries of the rows longer than k only, in lane
    // order; slice_base counts entries, col16 is indexed like val (no quads), slice_base16 is unused
    int jag;
    // x window (jagged slices only): the columns a 256-row group touches, merged into runs of consecutive columns,
    // are copied into LDS once per group with coalesced loads; col16 then holds 16-bit LDS SLOTS instead of offsets
    // (10 bytes per non-zero whatever the bandwidth of the matrix) and the x gather is an LDS read.
    // win_ptr[g] .. win_ptr[g+1]: the group's runs in win_runs, each {first column, (first slot << 16) | length}
    const uint32_t *win_ptr;
    const uint2    *win_runs;
    uint32_t        win_slots;   // LDS doubles the largest group needs (0: no window)
    // with windows the x gather is an LDS read, so lanes need not hold CONSECUTIVE rows any more: the rows of a
    // 256-row group are dealt to the lanes by decreasing length (perm[g * 256 + lane] = row within the group),
    // which makes the four slices of a group as long as their own longest row instead of the group's (null: lane = row)
    const unsigned char *perm;
    // ... and, for the product with the short dependency chain (bicg_jagw.hip), ONE 16-bit word per lane: its row within the group
    // (low byte, = perm) and that row's length (high byte) -- no row pointers, no separate permutation load. win_max_runs: the most
    // runs any group's window has. Null when some row of a sliced group has more than 255 entries.
    const unsigned short *lane_info;
    uint32_t win_max_runs;
    // List-driven window (one rank, round 6): the group's distinct columns one by one -- for numberings whose groups touch many
    // SHORT runs (reverse Cuthill-McKee of a tetrahedral mesh: 59-170 runs of ~25 columns, too many descriptors to search per slot).
    // 16 bits per column (distance from the group's first row), two slots per word: word win_list[win_lptr[g] + 256 j + t] holds
    // the columns of slots t + 512 j (low half) and t + 512 j + 256 -- the slots thread t stages; win_ltotal[g] slots in all.
    // k_spmv_jagw<.., LIST> loads its slots' columns instead of searching the runs; the runs (merged with gap 0) stay for
    // k_spmv_sell's loop. Null: no list.
    const uint32_t *win_list;
    const uint32_t *win_lptr;
    const uint32_t *win_ltotal;
    // Uniform slices (padded layouts): when all 64 rows of a slice are present, equally long and entry k of every row sits at
    // the SAME distance from its row -- every interior slice of a banded or stencil matrix -- the slice's columns are the list
    // uoff[ubase[slice] + k] (shared by all slices with the same list) and the SpMV does not read its col / col16 entries at
    // all: 8 instead of 10 (12) bytes per non-zero, the offsets arrive as scalar loads. ubase[slice] = 0xFFFFFFFF: not uniform.
    // (col / col16 stay complete: the SpMM and the window-fused product read them.)
    const uint32_t *ubase;
    const int      *uoff;
    // Constant slices: a uniform slice whose 64 rows also hold the SAME VALUE in entry k (the interior of a constant-coefficient
    // stencil, e.g. the 7-point Laplacian of BASELINE.json configs[3]): its values are the list uval[vbase[slice] + k], scalar
    // loads like the distances -- the slice streams NOTHING from the matrix arrays. vbase[slice] = 0xFFFFFFFF: values from val.
    const uint32_t *vbase;
    const double   *uval;
    // Masked slices: the 64 rows of a slice next to a grid face are NOT equally long -- a row on the face lacks the neighbour
    // beyond it -- but every row is a sub-sequence of one list of (distance, value) pairs (at most 16, ascending columns, equal
    // values where present). Such a slice keeps that list in uoff / uval like a constant slice and ONE 16-bit word per row, the
    // set of list entries the row has: its product reads 2 bytes per row from the matrix side instead of 10-12 per entry, and
    // adds the present entries in list = stored order. mbase[slice] = list length << 26 | index of the slice in rmask (64 words
    // per slice), 0xFFFFFFFF: not masked. With this the whole 7-point Laplacian of BASELINE.json configs[3] streams no values
    // and no columns at all.
    const uint32_t *mbase;
    const unsigned short *rmask;
    // One descriptor per slice for the blocks that have list-driven slices (PAD32C / PAD16C), so that a constant or masked slice
    // costs ONE scalar load of metadata instead of five from five arrays, and the product can request the next group's
    // descriptor while it multiplies the current one (a 7-entry slice of the Laplacian is three dependent round trips otherwise):
    //   x = length | kind << 16   (kSliceGeneral: columns and values streamed, kSliceUniform: values streamed,
    //                               kSliceConstant: y = position in uoff, z = position in uval,
    //                               kSliceMasked: the same + w = index of the slice in rmask; length = the LIST's length)
    // null: no descriptors (BICG_PLAN="desc=0", or 2^29 rows and more: the fast paths address x by 32-bit byte offsets).
    const uint4 *sdesc;
    // whole 256-row groups only, and every slice is constant or masked with a list of at most 8 entries (a constant-coefficient stencil: the
    // 7-point Laplacian of BASELINE.json configs[3]): the product runs a loop of its own over the descriptors -- list lengths
    // as compile-time cases, the distances as byte offsets (uoff8[i] = 8 uoff[i]) added to the row's 32-bit byte offset
    int all_lists;
    const int *uoff8;
    int ystride;            // all_lists: slices per grid line when the lists look like a grid's (second-largest distance 64 x a power of
                            // two rows, slices a multiple of 4 x that), else 0 -- the four wavefronts of a workgroup take slices this far apart
    StencilDev st;          // all_lists and every list a sub-sequence of (-sz, -sy, -1, 0, +1, +sy, +sz): the plane-marching product
};
enum SliceKind { kSliceGeneral = 0, kSliceUniform = 1, kSliceConstant = 2, kSliceMasked = 3 };
// (PAD32C / PAD16C: padded slices of a block that has CONSTANT slices -- SellDev::vbase. Instantiations of their own: with the
// value list as a run-time branch in every kernel, the dot-carrying products of blocks WITHOUT such slices paid 11 us each for
// the registers it took, 45 -> 56 us on Transport)
enum SellLayout { LAY_PAD32 = 0, LAY_PAD16 = 1, LAY_JAG32 = 2, LAY_JAG16 = 3, LAY_JAGW = 4, LAY_PAD32C = 6, LAY_PAD16C = 7 };
constexpr uint32_t kWinMaxSlots = 4096;      // 32 KB of LDS per workgroup: 4 workgroups per CU
constexpr uint32_t kJagwMaxRuns = 64;        // k_spmv_jagw: run descriptors one wavefront holds (lane r = run r) ...
constexpr uint32_t kJagwMaxSlots = 2048;     // ... and window values it stages (8 per thread)

// Peer-to-peer halo exchange folded into the sliced-ELL SpMV launch: the first `npush` workgroups
// store this rank's send list into the landing rings of the ranks that need it, the others
// multiply; offd entries read their x value straight from this rank's landing ring (LL words of
// exchange `seq`), spinning until it has arrived.
struct HaloLL {
    const llword *ring;                 // this rank's landing ring [kHaloRing][halo][2]
    uint32_t halo;
    unsigned seq;
    unsigned npush;                     // leading workgroups that push instead of multiplying
    uint32_t first_bnd;                 // list positions >= first_bnd hold halo-touching groups (only they have offd entries)
    uint32_t nsend;
    const uint32_t *send_idx;           // [nsend] local rows to send
    const unsigned long long *dst0;     // [nsend] address of the word pair in slot 0 of the receiver's ring
    const unsigned long long *dstride;  // [nsend] bytes per slot of the receiver's ring
    unsigned long long timeout_ticks;
};

// element-wise phase kernels: pointers to the rank-local vectors
struct Vecs {
    double *x, *r, *rh, *p, *s, *y, *z, *w, *v, *t, *ax, *b;   // shifted solver: ax doubles as r_old
    uint32_t n;
};

// Clusters of column distances of a padded 16-bit block: the columns of a 256-row group are g0 + row-in-group + d with d from a
// small set of offsets that falls into a few clusters (Transport: {-13807..-13689}, {-118..118}, {13689..13807}); cluster k of every
// group is the window run [g0 + lo_k, g0 + 255 + hi_k], so the LDS slot of an entry is  thread + d + bias_k  -- no per-group plan.
// Used by the SpMM kernels (k_spmm_win MODE 0, k_spmm_pipe). (Round 3's window-fused product of plain BiCGStab, which formed q and p
// in such windows, was a negative result and left the tree in round 6: profiles/NOTES.md.)
constexpr int kFwMaxClusters = 4;
struct FusedWindow {
    int ncl;                         // clusters of column offsets, ascending (0: none)
    int lo[kFwMaxClusters], hi[kFwMaxClusters], bias[kFwMaxClusters];   // bias_k = slot0_k - lo_k
    unsigned slots;                  // LDS doubles of a window
};

struct SpmvArgs {
    SellDev sell;
    const uint32_t *glist;  // SELL launch: 256-row groups to process (null = groups 0..nlist-1)
    uint32_t nrows;         // local rows
    CsrDev diag;            // local columns
    const short *diag_col16;  // CSR-order 16-bit column offsets (col - row) for the rows-over-lanes kernel, or null
    int     rowsplit;       // the CSR row blocks of this context go to k_spmv_rows (a row is spread over T lanes)
    CsrDev offd;            // columns renumbered to rows + halo position; ptr over ALL local rows
    const uint4 *desc;      // row blocks of this launch: {first row, end row, first nnz, end nnz}
    uint32_t nlist;         // number of row blocks to process
    const double *x;        // [rows + halo]
    double       *y;        // [rows]
    const double *u;        // dot operand (NDOT >= 1): d0 = sum u_i y_i ; NDOT == 2 adds d1 = sum y_i^2 ;
                            // NDOT == 3: d0 = sum u_i y_i, d1 = sum u_i^2
    double  shift;          // y = A x + shift * x (shifted solver: A + sigma[seed] I, src/shifted_solver.c:259-260)
    int     has_shift;
    Scal   *S;
    Reduce  red;
    int     nt;             // stream the matrix arrays with non-temporal loads (Infinity-Cache policy)
    int     groups_per_wg;  // sliced-ELL: 256-row groups handled by one workgroup
    int     xcd_map;        // sliced-ELL: XCD-contiguous order of the groups (workgroup b runs on XCD b % 8; measurement knob)
    int     reverse;        // sliced-ELL: workgroup b takes group nlist - 1 - b. Consecutive products of a solve alternate
                            // direction, so each starts on the part of the matrix the previous one left in the Infinity Cache
    HaloLL  ll;             // launch_spmv_sell(..., fused_halo = true) only
    Finish  fin;            // a dot group of earlier kernels to finish in this launch (seq 0: none)
    Vecs    epi;            // launch_spmv_sell_epi: the vectors of the element-wise phase in the epilogue
};

// ---- persistent pipelined iteration for latency-bound ranks (bicg_persist.hip) ---------------------------------
// ONE launch runs `niter` iterations of pipe_bicgstab (reference src/solver.c:351-398). A workgroup of 64 * spw
// threads owns spw consecutive 64-row slices for the whole launch (lane = row): its rows' vectors live in registers,
// its part of the matrix and the x values its rows touch live in LDS. Nothing crosses a kernel boundary, so whatever
// one workgroup needs from another travels as LL words (8 bytes = 32 payload bits + 32-bit sequence tag, one store, a
// stale word is recognisable -- no flags, no fences): the two SpMV input vectors z and w (every row publishes its
// value, the consumers' window loads spin on the tags), the dot partials (one row of a table per workgroup) and the
// applied scalars (one row, written by a helper workgroup that owns no rows: it adds the table in a fixed order,
// exchanges the sums with the other ranks through the peer-to-peer mailboxes, applies the recurrence and publishes
// alpha / beta / omega / done). Halo values of other ranks arrive in the landing ring as LL words exactly as in the
// multi-launch path and are read by the same window loads.
struct PersistArgs {
    uint32_t nrows, nslices, nwg, spw;   // nwg row workgroups (+ 1 helper) of 64 * (spw + 1) threads: spw row wavefronts ...
    uint32_t rpt;                        // ... whose threads own rpt rows each: a workgroup holds spw * rpt consecutive slices
    // matrix, padded slices with diag entries first, then offd entries (x_ext numbering): entry k of lane l of slice s
    // at pbase[s] + k * 64 + l; pslot = slot of the entry's column in the workgroup's window
    const double         *pval;
    const unsigned short *pslot;
    const uint32_t       *pbase;         // [nslices + 1]
    const unsigned short *rlen, *rdiag;  // [nrows] entries of the row / of its diag part
    const uint32_t *win_ptr;             // [nwg + 1] runs of workgroup g
    const uint2    *win_runs;            // {first column (>= nrows: halo position + nrows), (first slot << 16) | length}
    uint32_t win_slots, max_runs;        // LDS doubles of the largest window; most runs of one workgroup
    uint32_t mat_entries;                // > 0: the workgroup's matrix entries are copied into LDS (at most this many)
    llword *llv[4];                      // [nrows][2] local LL images of the vectors handed over (2 and 3: replacement iterations)
    llword *dtab[2];                     // [nwg][kRedSlots][2] dot partials of the two groups
    llword *arow[2];                     // [4][2] applied scalars after each group: alpha, beta, omega, done
    unsigned seq0;                       // table tags of this launch: group g (1, 2, ...) carries seq0 + g
    unsigned vseq0;                      // pipelined kernel: hand-off n (1, 2, ...) carries image tag vseq0 + n
    int niter;
    // pipelined kernel: residual replacement (src/solver.c:494-548) inside the launch
    int it0;                             // iterations completed before this launch (the reference's k)
    int krr, nrr;                        // pipe_bicgstab_rr's schedule: k % krr == 0, 0 < k <= krr nrr (krr = 0: none)
    int force_first;                     // the first iteration of this launch is a replacement iteration
    int drift_every;                     // > 0: every drift_every iterations compare b - A x with r (adaptive replacement) ...
    double drift_tol2;                   // ... and replace when ||(b - A x) - r||^2 > drift_tol2 ||r||^2
    // halo (multi rank, peer-to-peer transport)
    int multi;
    const llword *ring; uint32_t halo; unsigned halo_seq0;        // exchange numbers halo_seq0 + 1, + 2, ...
    const uint32_t *snd_ptr;             // [nwg + 1] send-list entries of workgroup g
    const unsigned short *snd_row;       // row within the workgroup
    const unsigned long long *snd_dst0, *snd_stride;
    P2pRed p2p;                          // mailboxes; group numbers p2p.seq + 2 it / + 1
    // shifted pipelined kernel (k_shpipe_persist): v.x / v.p are x[seed] / p[seed]; every other shift's x_j, p_j are streamed
    // through in phase 2 with the coefficients the helper publishes as LL pairs together with omega
    double *pset, *xset;                 // [nsig][set_stride]
    uint32_t set_stride;
    int nsig, seed;
    int set_nt;                          // stream the sets past the Infinity Cache (they do not fit it)
    double shift; int has_shift;         // products are (A + shift I) x                              (src/shifted_solver.c:259-260)
    llword *crow[2];                     // [6][kPersistMaxShifts][2] beta_j, alpha_j, cp, cx, c1, c2 of the group's iteration
    Vecs v;
    Scal *S;
    int *alarm;
    unsigned long long timeout_ticks;
    unsigned first_sleep;                // row wavefronts: s_sleep(8) periods (~0.22 us each) before the first look at the window
    int xcd_map;                         // XCD-contiguous assignment of row ranges to workgroups
    unsigned long long *dbg;             // BICG_PERSIST_TRACE: 100 MHz time stamps of one row workgroup and the helper, [it][16]
    // multi-rank launches: how long the exchanges made the kernel wait, in 100 MHz ticks, one sample per exchange --
    // row 0: the helper's all-reduce through the mailboxes (own sums stored -> every rank's sums read), index = group % waitcap;
    // rows 1 / 2: first / last row workgroup, own values published -> window complete (the neighbours' halo values included),
    // index = hand-off % waitcap. Read back by bicg_comm_wait_stats (bench.py's comm.wait_us). Null: not recorded.
    unsigned *waitlog;
    unsigned waitcap;
};
// each returns hipSuccess or the reason the launch did not happen (launch error, workgroups cannot be co-resident): the
// caller then runs the chunk with the multi-launch kernels
hipError_t launch_pipe_persist(const PersistArgs &a, hipStream_t st);
hipError_t launch_plain_persist(const PersistArgs &a, hipStream_t st);    // plain BiCGStab: three groups per iteration
hipError_t launch_ca_persist(const PersistArgs &a, hipStream_t st);       // CA-BiCGStab: two groups per iteration
hipError_t launch_shpipe_persist(const PersistArgs &a, hipStream_t st);   // shifted_pipe_lopbicgstab (src/shifted_solver.c:794-866), <= kPersistMaxShifts shifts
hipError_t launch_shlop_persist(const PersistArgs &a, hipStream_t st);    // shifted_lopbicgstab (src/shifted_solver.c:257-319), same limits
constexpr int kPersistMaxShifts = 32;
unsigned persist_lds_bytes(const PersistArgs &a);
constexpr unsigned kPersistMaxLds = 160u * 1024u - 1024u;     // dynamic LDS a launch may ask for (static part: < 1 KiB)

// Sliced-ELL SpMM over kSpmmCols vectors held row-major (bicg_kernels.hip, k_spmm_sell)
constexpr int kSpmmCols = 16;
struct SpmmArgs {
    SellDev sell;
    const uint32_t *dptr;   // diag row pointers (row lengths)
    CsrDev offd;            // columns renumbered to rows + halo position
    uint32_t nrows, ngroups;
    const double *xt;       // [rows + halo][kSpmmCols]
    double *yt;             // [rows][kSpmmCols] or null
    const double *b;        // with b: partial[wg][col] = sum over the workgroup's rows of (b_i - y_ij)^2
    const double *sigma;    // [kSpmmCols] or null: y_j += sigma_j x_j
    double *partial;
    int xcd_map;            // XCD-contiguous assignment of row groups
    // windowed form (k_spmm_win): the vectors stay SHIFT-MAJOR (x_j = xs + j * vstride, halo tails filled); the x values a
    // 256-row group touches are staged in LDS for NV vectors at a time -- from the cluster runs of padded 16-bit layouts
    // (cl.ncl > 0, struct FusedWindow) or from the window runs of the layout with LDS slots (sell.win_runs)
    const double *xs;
    double *ys;             // [nvec][vstride] or null
    size_t vstride;
    int nvec;
    FusedWindow cl;
    unsigned wslots;        // LDS doubles per vector
    int dbg;                // BICG_TEST="spmm-skip=n" (measurement only, results are wrong): 1 no staging loads, 2 no products, 4 no row heads
    unsigned tail_most;     // k_spmm_jpipe: most entries one jagged slice holds behind the 16th entry of its rows (the plan's count)
    double gstep;           // k_spmm_pipe: groups per workgroup (fractional: workgroup w takes groups floor(w gstep) .. floor((w + 1) gstep) - 1)
};


// which product kernels have been launched since the last reset (bicg_product_kernels: tests and bench.py assert on the kernel
// a matrix gets, not only on the plan's flags)
enum ProductKernel : unsigned { PK_SELL_PAD = 1, PK_SELL_JAG = 2, PK_SELL_WINLOOP = 4, PK_JAGW = 8, PK_STENCIL = 16, PK_CSR = 32, PK_ROWS = 64,
                                PK_SELL_EPI = 128, PK_JAGD = 512, PK_JAGW_LIST = 1024 };
extern unsigned g_product_kernels;

// ---- launch wrappers (bicg_kernels.hip) ----
// Both return false when there was nothing to launch. e0/e1 (optional): start/stop events bound to
// this one kernel (hipExtLaunchKernelGGL) -- the per-kernel durations bench.py's roofline uses.
bool launch_spmv(const SpmvArgs &a, int ndot, bool with_offd, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);        // CSR row-block stream
bool launch_spmv_sell(const SpmvArgs &a, int ndot, bool with_offd, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                      bool fused_halo = false);   // sliced ELL; fused_halo: a.ll describes the in-kernel exchange
// plane-marching product of a 7-point grid stencil (bicg_stencil.hip; a.sell.st.on). epi = 1: CA-BiCGStab's q = r - alpha s,
// y = w - alpha z, (q,y), (y,y) (reference src/solver.c:225-232) on the wavefront's own rows behind z = A s (a.epi.r / a.epi.w)
// the ragged-rows product with three dependent trips per group (bicg_jagw.hip); jagw_fast_ok: this launch qualifies
bool jagw_fast_ok(const SpmvArgs &a, bool with_offd, bool fused_halo);
bool launch_spmv_jagw(const SpmvArgs &a, int ndot, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// ... and its form for jagged slices WITHOUT a window (x gathered through the caches: 16-bit offsets or 32-bit columns)
bool jagd_fast_ok(const SpmvArgs &a, bool with_offd, bool fused_halo);
bool launch_spmv_jagd(const SpmvArgs &a, int ndot, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void preload_jagw_kernels();
unsigned stencil_grid(const StencilDev &st);
bool launch_spmv_stencil(const SpmvArgs &a, int ndot, int epi, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void preload_stencil_kernels();
void launch_spmm_sell(const SpmmArgs &a, bool with_offd, hipStream_t st);
// vectors per LDS window of the windowed form for `wslots` doubles per vector (0: the window does not fit, use launch_spmm_sell)
int spmm_win_vectors(unsigned wslots);
hipError_t launch_spmm_win(const SpmmArgs &a, bool with_offd, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);   // e0 / e1: stamped at the kernel's start / end
// pipelined form (bicg_spmm.hip, k_spmm_pipe): the window of the next step copied global -> LDS by the DMA path while the current one
// multiplies, persistent workgroups over consecutive groups; padded 16-bit layouts with clusters (hipErrorInvalidValue: not this block)
hipError_t launch_spmm_pipe(const SpmmArgs &a, bool with_offd, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// the same pipeline on jagged slices with x windows (bicg_spmm_jag.hip, k_spmm_jpipe); hipErrorInvalidValue: the block does not qualify
hipError_t launch_spmm_jpipe(const SpmmArgs &a, bool with_offd, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void preload_spmm_kernels();
unsigned spmm_grid(uint32_t ngroups, bool xcd_map);
// look up one kernel of every translation unit a context with this sliced-ELL plan launches from (loads their code objects now)
void preload_kernels(const SellDev &d, bool sell);
void preload_persist_kernels();
void launch_colsum(const double *partial, unsigned nwg, double *out, hipStream_t st);          // out[col] = sum_wg partial[wg][col]
void launch_rows_from_vectors(const double *x, size_t stride, int nvec, uint32_t n, double *xt, hipStream_t st);
void launch_vectors_from_rows(const double *yt, size_t stride, int nvec, uint32_t n, double *y, hipStream_t st);
// SpMV + pipelined phase in the epilogue (epi 1: phase 2 after v = A z; epi 2: phase 1 after t = A w); a.fin is
// applied at the epilogue, a.red receives the phase's dot partials
bool launch_spmv_sell_epi(const SpmvArgs &a, int epi, bool with_offd, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                          bool fused_halo = false);
void launch_apply(Scal *S, int phase, hipStream_t st);
void launch_finish(const Launch &L);   // stand-alone finisher: L.fin with FIN_BLOCK0
void launch_halo_pack(const double *x, const uint32_t *send_idx, uint32_t nsend, double *sendbuf, Scal *S, hipStream_t st);
// peer-to-peer transport: wait for the P contributions of group pr.seq, sum n values, apply `phase`
void launch_apply_p2p(Scal *S, int phase, int n, const P2pRed &pr, unsigned long long timeout_ticks, hipStream_t st);
// halo exchange: entry i of the send list goes to dst0[i] + (seq % kHaloRing) * dst_stride[i] bytes
// (a word pair in the landing ring of the rank that needs it); the receiver decodes its ring slot
// into the halo tail of the vector
void launch_halo_push(const double *x, const uint32_t *send_idx, uint32_t nsend, const unsigned long long *dst0,
                      const unsigned long long *dst_stride, unsigned seq, Scal *S, hipStream_t st);
void launch_halo_unpack(const llword *ring, uint32_t halo, unsigned seq, double *tail, Scal *S,
                        unsigned long long timeout_ticks, hipStream_t st);
// barrier token pr.seq (own sequence space) among all ranks
void launch_p2p_barrier(const P2pRed &pr, unsigned long long timeout_ticks, Scal *S, hipStream_t st);
// transport self-test: `rounds` all-reduces of known values starting at sequence seq0; status[0]
// counts mismatches, status[1] time-outs
void launch_p2p_selftest(const P2pRed &pr, unsigned seq0, int rounds, unsigned long long timeout_ticks, int *status,
                         hipStream_t st);
// ... and of the halo pattern: `rounds` exchanges of `entries` values per rank pair through test rings laid out
// [kHaloRing][source rank][entries][2], slots reused, ranks out of step, token barrier (sequence bar_seq0...) every
// kHaloRing - 2 rounds; status[0] counts wrong / stale values, status[1] time-outs
void launch_p2p_ringtest(const P2pRed &pr, llword *const *rings, int entries, unsigned seq0, int rounds, unsigned bar_seq0,
                         unsigned long long timeout_ticks, int *status, hipStream_t st);

// The kernels of the four solvers take a Launch: the scalar block to read, the dot group of earlier
// kernels to finish first (if any) and the stream.
// init: r = b - Ax ; rh = r ; [p = r] ; [bsave = b] ; dot (r,r)
void launch_init_residual(const Vecs &v, bool copy_p, bool save_b, const Launch &L, Reduce red);
// plain BiCGStab phases (src/solver.c:94, 105-111, 117-119)
void launch_plain_q(const Vecs &v, const Launch &L);
void launch_plain_xr(const Vecs &v, const Launch &L, Reduce red, const double *q = nullptr);   // q: where q lives (default: in r)
void launch_plain_p(const Vecs &v, const Launch &L);
// CA-BiCGStab phases (src/solver.c:217-222, 225-228, 233-236 + 240-243)
void launch_ca_ps(const Vecs &v, const Launch &L);
void launch_qy(const Vecs &v, const Launch &L, Reduce red);
void launch_ca_xr(const Vecs &v, const Launch &L, Reduce red);
// pipelined phases (src/solver.c:352-364, 370-380)
void launch_pipe_f1(const Vecs &v, const Launch &L, Reduce red);
void launch_pipe_f2(const Vecs &v, const Launch &L, Reduce red);
// residual-replacement steps (src/solver.c:494-496, 519-520, 524-525, 533-538)
void launch_p_update(const Vecs &v, const Launch &L);
void launch_x_update(const Vecs &v, const Launch &L);
void launch_true_residual(const Vecs &v, const Launch &L);
void launch_dots5(const Vecs &v, const Launch &L, Reduce red);
// shifted BiCGStab (reference src/shifted_solver.c:182-354)
void launch_shift_init(const Vecs &v, double *p_seed, Scal *S, Reduce red, hipStream_t st);      // r# = r, p[seed] = r, (r,r)
void launch_shift_q(const Vecs &v, Scal *S, hipStream_t st);                                     // r_old = r ; q = r - alpha s
// x[seed], r, the two dots, and for every other shift j: p_j and x_j (one pass over both sets)
void launch_shift_update(const Vecs &v, double *p_set, double *x_set, uint32_t set_stride, int seed, const ShiftDev *H,
                         Scal *S, Reduce red, hipStream_t st);
void launch_shift_pseed(const Vecs &v, double *p_seed, Scal *S, hipStream_t st);                 // p[seed] = r + beta (p[seed] - omega s)
// pipelined shifted variant: phase 1 (p[seed], s, z recurrences, r_old, q, y, 2 dots) and phase 2
// (x[seed], every p_j / x_j, r, w, 5 dots)
void launch_shift_pipe1(const Vecs &v, double *p_seed, Scal *S, Reduce red, hipStream_t st);
void launch_shift_pipe2(const Vecs &v, double *p_set, double *x_set, uint32_t set_stride, int seed, const ShiftDev *H,
                        Scal *S, Reduce red, hipStream_t st);
// seed-switching shifted solvers (reference src/shifted_switching_solver.c)
void launch_sw_q(const Vecs &v, double *qcopy, Scal *S, hipStream_t st);                  // r_old = r ; q = r - alpha s -> r, q_copy
// x[seed] += alpha p[seed] + omega q ; r = q - omega y ; (r,r), (r#,r)
void launch_sw_seed(const Vecs &v, double *x_seed, const double *p_seed, Scal *S, Reduce red, hipStream_t st);
// p[seed] = beta p[seed] + r - beta omega s, and for every active shift the x_j / p_j updates of the iteration
void launch_sw_shifts(const Vecs &v, const double *qcopy, double *p_set, double *x_set, uint32_t set_stride, int seed,
                      const ShiftDev *H, Scal *S, hipStream_t st);
void launch_scale(double *x, uint32_t n, double a, hipStream_t st);                        // x <- a x (my_dscal)
// adaptive residual replacement: red[0] = ||(b - Ax) - r||^2, red[1] = ||r||^2
void launch_drift(const Vecs &v, const Launch &L, Reduce red);
// standalone dot (x,y) -> red[0]
void launch_dot(const double *x, const double *y, uint32_t n, Scal *S, Reduce red, hipStream_t st);

// device-side sliced-ELL plan (bicg_plan_device.hip): slice lengths + "some column is further than 32767 from its row",
// then the column-major padded copy (32-bit columns or packed 16-bit offsets)
void launch_plan_rowstats(const uint32_t *ptr, const uint32_t *col, uint32_t rows, uint32_t *slice_len, int *far, hipStream_t st);
void launch_plan_uniform(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows, unsigned long long *uhash,
                         unsigned long long *vhash, hipStream_t st);
// masked slices (SellDev::mbase): mhash[s] = hash of the slice's list of (distance, value bits) with its length in the low 5 bits
// (0: not masked); second call with mbase given: the rows' masks of the masked slices into rmask
void launch_plan_masked(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows, unsigned long long *mhash,
                        const uint32_t *mbase, unsigned short *rmask, hipStream_t st);
void launch_plan_verify(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows, const uint32_t *slice_len,
                        const uint32_t *ubase, const uint32_t *vbase, const uint32_t *mbase, const unsigned short *rmask, const int *uoff,
                        const double *uval, unsigned char *bad, hipStream_t st);
void launch_plan_fill(const uint32_t *ptr, const uint32_t *col, const double *val, uint32_t rows, const uint32_t *slice_base,
                      const uint32_t *slice_base16, double *sval, uint32_t *scol, short *scol16, hipStream_t st);

unsigned sell_grid(uint32_t ngroups, int per_wg); // workgroups launched for ngroups 256-row groups
unsigned vec_grid(uint32_t n);        // workgroups used by the element-wise kernels for length n
void set_vec_grid_cap(unsigned cap);  // ranks sharing one GPU (tests): fewer workgroups per launch (0 = default)
unsigned spmv_grid(uint32_t nlist);   // workgroups used by the CSR SpMV for nlist row blocks

}  // namespace bicg
