// bicg_host.h -- what the host-side translation units of the library share: the context of a rank (struct bicg_ctx), small
// device-memory helpers, section timing, and the prototypes of the functions that cross file boundaries.
//   bicg_solver.cpp   dot groups, the distributed SpMV, the four iterations of reference src/solver.c, run_begin / iterate / end
//   bicg_shifted.cpp  the shifted family (src/shifted_solver.c, src/shifted_switching_solver.c) and its section prints
//   bicg_create.cpp   the plan: bicg_create / bicg_create_device_csr, slice descriptors, stencil plan, persistent set-up, destroy
//   bicg_dropin.cpp   the reference's own symbols (solver.h, shifted_solver.h, shifted_switching_solver.h), matrix residency
//   bicg_api.cpp      the additive handle API (load / fetch / spmv / dot / spmm / info calls)
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "bicg_comm.h"
#include "bicg_knobs.h"
#include "bicg_plan.h"
#include "bicg_parallel.h"
#include <memory>
#include "bicg_device.h"

using namespace bicg;



constexpr int kEvRing = 16;
constexpr int kMaxTimed = 8192;
constexpr int kPersistChunk = 128;   // iterations per persistent launch, at least (run_iterate)

inline double now_sec()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class T> T *dev_alloc(size_t n)
{
    T *p = nullptr;
    BICG_HIP(hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)));
    return p;
}

template <class T> T *dev_upload(const T *src, size_t n)
{
    T *p = dev_alloc<T>(n);
    if (n) BICG_HIP(hipMemcpy(p, src, sizeof(T) * n, hipMemcpyHostToDevice));
    return p;
}

// n entries followed by `pad` zero entries (16-byte loads may run past the last non-zero)
template <class T> T *dev_upload_padded(const T *src, size_t n, size_t pad)
{
    T *p = dev_alloc<T>(n + pad);
    BICG_HIP(hipMemset(p + n, 0, sizeof(T) * pad));
    if (n) BICG_HIP(hipMemcpy(p, src, sizeof(T) * n, hipMemcpyHostToDevice));
    return p;
}


constexpr unsigned kWaitCap = 4096;      // samples per row of PersistArgs::waitlog
struct bicg_ctx {
    Comm *comm = nullptr;                  // null once the communicator has been replaced (contexts_orphan)
    int device = 0;
    int nranks = 1, rank = 0;
    uint32_t n_loc = 0, n_glob = 0, halo = 0, stride = 0, nnz_d = 0, nnz_o = 0;

    // matrix + plan (device)
    double *d_val = nullptr, *o_val = nullptr;
    uint32_t *d_col = nullptr, *d_ptr = nullptr, *o_col = nullptr, *o_ptr = nullptr;
    uint4 *desc_int = nullptr, *desc_bnd = nullptr;   // CSR row-block descriptors: interior / halo-touching
    FusedWindow fw{};                      // clusters of column distances of a padded 16-bit block (fw.ncl > 0): the windows of the SpMM kernels
    bool rowsplit = false;                 // long rows: the row blocks go to k_spmv_rows (a row spread over T lanes)
    short *d_col16 = nullptr;              // ... with CSR-order 16-bit column offsets when they fit
    uint32_t nblk = 0, n_int = 0, n_bnd = 0;
    int sell_gpw = 1, sell_gpw_dots = 1;   // 256-row groups per workgroup: plain SpMV / SpMV with fused dots
    uint32_t sell_blocked = 0;             // the groups are taken plane block by plane block (sell_order_for_big_grids): block size
    int spmv_dir = 0;                      // direction of the last sliced-ELL product (SpmvArgs::reverse)
    int sell_alt = 1;                      // BICG_SELL_ALT=0: every product forward; default: consecutive products alternate direction
    int sell_xcd = 1;                      // BICG_SELL_XCD=0: round robin; default: XCD-contiguous group order (SpmvArgs::xcd_map)
    int sell_nt_env = -1;                  // BICG_SELL_NT: force (1) / forbid (0) non-temporal matrix loads
    bool sell_nt = false;                  // decided per solve from the working-set size (run_begin)
    uint64_t matrix_bytes = 0;             // bytes one SpMV streams from the matrix arrays
    uint64_t stencil_matrix_bytes = 0;     // ... when the plane-marching product runs (StencilDev)
    hipEvent_t region_ev[2] = {nullptr, nullptr};   // bicg_run_iterate_timed
    unsigned *waitlog = nullptr;           // PersistArgs::waitlog (multi-rank persistent launches), 3 rows of kWaitCap samples
    double t_enq = 0.0;
    uint64_t device_matrix_bytes = 0;      // bytes of matrix storage resident on the GPU
    // sliced-ELL copy of the diag block (rows whose 256-row group pads by < 25 %)
    double *s_val = nullptr;
    uint32_t *s_col = nullptr, *s_base = nullptr, *s_len = nullptr, *s_base16 = nullptr;
    short *s_col16 = nullptr;
    uint32_t *s_ubase = nullptr;           // uniform slices (SellDev::ubase / uoff): BICG_PLAN="uniform=0" switches them off
    int *s_uoff = nullptr;
    uint64_t uniform_entries = 0;          // sliced-ELL entries whose columns the SpMV does not read
    uint32_t far_rows = 0;                 // farthest column distance of a uniform slice, in rows (a grid's plane size)
    uint32_t *s_mbase = nullptr;           // masked slices (SellDev::mbase / rmask): BICG_PLAN="masked=0" switches them off
    unsigned short *s_rmask = nullptr;
    uint64_t masked_rows = 0;
    uint32_t plan_collisions = 0;          // list-driven slices the device plan's verification pass put back (bicg_plan_collisions)
    int *s_uoff8 = nullptr;                // SellDev::uoff8
    int sell_ystride = 0;                  // SellDev::ystride (BICG_SELL_YGROUP=1; default: consecutive slices per workgroup)
    bool sell_all_lists = false;           // SellDev::all_lists (BICG_PLAN="lists=0" switches the loop of its own off)
    StencilDev st{};                       // SellDev::st: the plane-marching product of a 7-point grid stencil (BICG_PLAN="stencil=0": off)
    uint32_t *st_code = nullptr; StencilTab *st_tab = nullptr; unsigned char *st_cmask = nullptr; uint32_t *st_wbits = nullptr;
    bool st_multi = false;                 // several ranks: the halo-free rows of this rank are the whole planes z_lo .. z_hi - 1 of its grid
    bool ca_fuse = true;                   // CA-BiCGStab: q, y and their dots in the epilogue of z = A s (plane-marching product only; BICG_PLAN="ca-fuse=0")
    uint4 *s_desc = nullptr;               // one descriptor per slice (SellDev::sdesc): BICG_PLAN="desc=0" switches them off
    uint32_t *s_vbase = nullptr;           // constant slices (SellDev::vbase / uval): BICG_PLAN="constant=0" switches them off
    double *s_uval = nullptr;
    uint64_t constant_entries = 0;         // ... whose values it does not read either
    bool sell_jag = false;                 // jagged slices (ragged rows: no padding stored), SellDev::jag
    uint32_t *win_ptr = nullptr, win_slots = 0;   // x windows in LDS (SellDev::win_*)
    uint32_t win_max_runs = 0;             // most runs of one group's window
    uint32_t jag_tail16_max = 0;           // most entries one jagged slice holds behind the 16th entry of its rows
    bool win_near16 = false;               // every window column lies within 16 bits of its group's first row (k_spmm_jpipe packs them)
    uint2 *win_runs = nullptr;
    unsigned char *sell_perm = nullptr;    // SellDev::perm
    unsigned short *lane_info = nullptr;   // SellDev::lane_info
    uint32_t *win_list = nullptr, *win_lptr = nullptr, *win_ltotal = nullptr;     // list-driven window (SellDev::win_list)
    bool jagw_fast = true;                 // the three-trip product of bicg_jagw.hip (BICG_PLAN="jagw=0": k_spmv_sell's loop)
    uint32_t *glist_int = nullptr, *glist_bnd = nullptr;
    uint32_t ng_int = 0, ng_bnd = 0, sell_rows = 0;
    uint64_t sell_entries = 0, sell_nnz = 0;
    bool glist_int_identity = false;
    bool glist_all = false;        // every 256-row group is on the sliced-ELL path (one merged launch possible)

    // halo exchange
    std::vector<int> scnt, sdsp, rcnt, rdsp;
    uint32_t nsend = 0;
    uint32_t *send_idx = nullptr;
    double *sendbuf = nullptr;
    // peer-to-peer transport (comm->p2p): landing ring for incoming halo values and, per entry of
    // the send list, where it goes in the ring of the rank that needs it
    P2p *p2p = nullptr;
    llword *halo_ring = nullptr;                              // [kHaloRing][halo][2]
    unsigned long long *push_dst0 = nullptr, *push_stride = nullptr;
    std::vector<void *> ring_mapped;
    unsigned halo_seq = 0;          // exchanges started (sequence number of the last one)
    int halo_unsynced = 0;          // exchanges since the last all-reduce or barrier (flow control)
    unsigned pend_seq = 0;
    bool comm_failed = false;       // a peer-to-peer wait timed out (BICG_P2P_SOFT_FAIL)
    bool soft_fail = false;         // ... report it through comm_failed instead of ending the program (drop-in fallback)
    // Exchange folded into the SpMV launch (HaloLL): possible when every halo-touching row is on the
    // sliced-ELL path. One launch covers push + interior + halo-touching groups (listed in that order).
    bool ll_fused = false;
    uint32_t *glist_ll = nullptr;
    bool inline_apply = true;       // BICG_P2P_INLINE_APPLY=0: always use the separate apply kernel
    int fault_after = 0;            // BICG_TEST="p2p-fault-after=n" (tests): from the n-th exchange on this rank sends nothing

    // vectors and scalars
    double *slab = nullptr;
    Vecs v{};
    Scal *S = nullptr;           // the scalar block kernels enqueued from now on read (= Sbuf + cur)
    Scal *Sbuf = nullptr;        // two blocks: a kernel that finishes a dot group reads one and writes the other
    int cur = 0;
    Scal *hS = nullptr;          // pinned mirror
    // consumer-side finish of dot groups (struct Finish, bicg_device.h): the four solvers of src/solver.c
    struct Group {
        bool active = false;     // produced, not yet consumed
        bool deferred = false;   // may ride across the next SpMV (pipelined variants, src/solver.c:363-367)
        bool staged = false;     // an SpMV launch has already summed the shards / pushed the sums to the peers
        unsigned seq = 0, mail_seq = 0, nparts = 0;
        int n = 0, off = 0, phase = 0, buf = 0;
    } grp;
    bool wave_mode = false;      // this call uses consumer-side finish (run_begin); false: ticket reductions
    bool spmm_ok = false;        // spmm_possible() on every rank (the SpMM exchanges the halos of all its vectors at once)
    bool fuse_plan_ok = false;   // every row on the sliced-ELL path and one launch per SpMV -- ON EVERY RANK (the fused and the
                                 // separate flow exchange their dot groups differently: the choice is collective)
    bool fuse_pipe = true;       // pipelined solvers: element-wise phases in the SpMV epilogues (BICG_PLAN="fuse-pipe=0|1" overrides)
    bool fuse_small = true;      // ... the average block has < 6 M non-zeros: fused whatever the layout
    int  pipe_probe = 0;         // BICG_PLAN="pipe-probe": the first pipelined solve TIMES both forms on this matrix and keeps the faster
    bool pipe_probed = false;    // ... done (the choice holds for the life of the context)
    double probe_ms[2] = {0, 0}; // ... ms per iteration measured for {separate kernels, phases in the SpMV epilogues}
    bool f1_done = false;        // phase 1 of the NEXT iteration has already run in the previous launch's epilogue
    // persistent pipelined iteration (bicg_persist.hip, struct PersistArgs): plan + LL buffers; persist.nwg == 0: not available
    PersistArgs persist{};
    bool persist_on = false;     // use it for pipe_bicgstab (every rank agrees); BICG_PERSIST=0 (or off) keeps the multi-launch iteration
    bool persist_plain = true;   // ... and for plain BiCGStab (BICG_PERSIST_PLAIN=0: the five-launch iteration)
    bool last_shifted_persist = false;   // the last shifted solve ran as persistent launches (bicg_result.flags of bicg_solve_shifted)
    unsigned persist_seq = 0;    // LL tags used so far (dot tables)
    unsigned persist_vseq = 0;   // ... by the pipelined kernel's vector images
    std::vector<void *> persist_mem;
    unsigned wg_cap = 0;         // ranks sharing this GPU (tests): workgroups per launch that may wait for another rank
    double *wpart[2] = {nullptr, nullptr};   // per-wavefront partial sums, alternating between groups
    llword *shard_ll = nullptr;  // 2 x [kShards][kRedSlots][2], alternating like wpart
    int *alarm = nullptr, *h_alarm = nullptr;
    unsigned grp_seq = 0;
    unsigned long long spin_ticks = 2000;   // 20 us before a workgroup sums a missing shard itself (BICG_TEST="spin-ticks=n")
    double *partial = nullptr, *shard_tot = nullptr;
    unsigned *counter = nullptr;
    // tail finish of ticket-mode dot groups (struct Reduce): LL table + shard totals; BICG_TAIL_FINISH=0: arrival tickets
    llword *tail_tab = nullptr, *tail_shard = nullptr;
    mutable unsigned tail_seq = 0;
    bool tail_finish = true;
    unsigned nslots = 0;
    double *trace = nullptr;     // 4 * trace_cap
    int trace_cap = 0;
    int last_iters = 0;

    hipStream_t sc = nullptr, sm = nullptr;   // compute, communication
    hipEvent_t ev_pack[kEvRing] = {}, ev_halo[kEvRing] = {}, ev_dots[kEvRing] = {}, ev_red[kEvRing] = {};
    unsigned i_pack = 0, i_halo = 0, i_dots = 0, i_red = 0;

    // deferred dot group (pipelined variant: all-reduce overlaps the next SpMV)
    bool pend = false;
    int pend_n = 0, pend_phase = 0, pend_off = 0;
    hipEvent_t pend_ev = nullptr;

    // shifted solver (bicg_solve_shifted): per-shift scalar state and the two vector sets
    double *sw_buf = nullptr;        // seed-switching variants: archives (doubles) followed by the flag arrays
    size_t sw_cap = 0;
    ShiftDev *sh_dev = nullptr;
    double *sh_arrays = nullptr, *p_set = nullptr, *x_set = nullptr;
    int sh_cap = 0;
    double cur_shift = 0.0;
    bool cur_has_shift = false;

    // SpMM (bicg_spmm, bicg_shifted_residuals): kSpmmCols shift-major vectors with halo tails, their row-major
    // image [rows + halo][kSpmmCols], the row-major result and the per-workgroup column sums
    double *mm_in = nullptr, *mm_xt = nullptr, *mm_yt = nullptr, *mm_part = nullptr, *mm_out = nullptr, *mm_sigma = nullptr;
    bool mm_xcd = true;          // XCD-contiguous row groups in the SpMM (BICG_SPMM_XCD=0: round robin like the SpMV)
    // A rank WITHOUT rows (more ranks than rows, or an empty block of a non-zero balanced partition; the reference's loops simply
    // run over zero rows there, src/matrix.c:295-298) holds ONE phantom row here -- the 1 x 1 block [1.0], decoupled from every
    // other row, with x = b = 0: all its vector entries stay 0, it adds 0.0 to every dot sum, sends and receives nothing, and so
    // takes part in every exchange and every launch path without a zero-row form of any kernel. The caller's vectors are empty:
    // host reads come from / host writes go to a scratch (host_in / host_out below).
    bool phantom = false;
    std::vector<double> ph_scratch;
    bool mm_win = false;         // the last SpMM pass ran the windowed kernel (vectors stay shift-major, X staged in LDS)
    int  mm_win_env = 3;         // BICG_PLAN="spmm-window=0": the row-major kernel, 1: k_spmm_win everywhere, 3 (default): k_spmm_pipe where the block qualifies
    bool mm_dma = false;         // the last SpMM pass ran the pipelined kernel (bicg_spmm.hip)

    // state of the solve in progress (run_begin / run_iterate / run_end)
    bicg_options opt{};
    int method = 0, it = 0, printed = 0, adaptive_rr = 0;
    double t_begin = 0.0, t_init = 0.0, t_iter = 0.0;

    // per-SpMV timing
    bool time_kernels = false;
    std::vector<hipEvent_t> tev;
    int tev_used = 0, spmv_calls_timed = 0;

    // section timing (bicg_options.time_kernels & 2): an event on the compute stream wherever the kind of work
    // changes; the time between two marks belongs to the section the first one opened. The counterpart of the
    // reference's MEASURE_SECTION_TIME (src/shifted_switching_solver.c:77-81, 132-154, 230-247: MPI_Wtime around the
    // shift loops, seed = total - shift), on the device's clock instead of the host's.
    bool time_sections = false, sec_exhausted = false;
    std::vector<hipEvent_t> sec_ev;
    std::vector<unsigned char> sec_lab;
    // finer attribution of a mark (the reference's ten sections, src/shifted_switching_solver.c:678-695): iteration it belongs to,
    // which product of the iteration (1 / 2), and what inside the product (0 the rows / everything, 1 halo exchange, 2 halo-touching rows)
    std::vector<int> sec_k;
    std::vector<unsigned char> sec_sub;
    int cur_k = 0, cur_prod = 0, cur_sub = 0;
    bool sec_dump = false;                 // BICG_SECTION_TIME=2: the per-iteration table of DISPLAY_SECTION_TIME
    double switch_sec = 0.0;               // host time spent in seed switches
    int sec_used = 0, sec_cur = 255;
    double sec_ms[4] = {0, 0, 0, 0};
    int sec_iters = 0;

    // BICG_TEST="force-comm" (tests): run the multi-rank code path (pack, exchange, packed all-reduce,
    // apply kernels, two streams) even with one rank, so that it can be exercised on a one-GPU box
    bool force_comm = false;
    bool single() const { return nranks == 1 && !force_comm; }

    // Use the second (communication) stream to overlap the halo exchange with the interior rows and
    // the pipelined variant's all-reduces with the next SpMV (reference src/matrix.c:432-440,
    // src/solver.c:363-367). A cross-stream hand-off costs ~7 us each way, the interior SpMV of a
    // 200 k-row rank only ~6 us, so below ~6 M local non-zeros everything is enqueued in order on
    // the compute stream instead. BICG_OVERLAP=0/1 overrides.
    bool overlap = false;

    // hipGraph replay of the iteration body (BICG_GRAPH): one captured iteration per method
    int graph_mode = -1;                 // -1 auto, 0 off, 1 on
    hipGraphExec_t graph_exec[4] = {nullptr, nullptr, nullptr, nullptr};
    int graph_warm[4] = {0, 0, 0, 0};    // eager iterations done since the context was created
    bool graph_nt[4] = {false, false, false, false};
    // now_n > 0: the group is closed by group_now(now_n, phase) right after this producer (not
    // deferred); with the peer-to-peer transport the producer's finishing workgroup then collects
    // and applies it in-kernel and group_now launches nothing.
    mutable bool open_inline = false;
    Reduce red(int off, int phase, bool apply_single = true, int now_n = 0) const
    {
        Reduce r{};
        r.partial = partial; r.shard_tot = shard_tot; r.counter = counter; r.expected = 0; r.slot_base = 0;
        r.red_off = off; r.phase = phase;
        r.apply_now = (single() && apply_single) ? 1 : 0;
        r.p2p = P2pRed{};
        r.tail_tab = tail_tab; r.tail_shard = tail_shard;
        // (not under hipGraph replay: a captured launch would meet its own earlier words under the same tag)
        r.tail_seq = (tail_finish && !p2p && tail_tab && graph_mode != 1) ? ++tail_seq : 0u;
        if (p2p) {
            r.p2p = p2p->red_desc(p2p->red_seq);   // the group being produced; closed by group_now/defer
            if (apply_single && now_n > 0 && inline_apply) {
                r.apply_now = 1; r.p2p.n_collect = now_n;
                open_inline = true;
            }
        }
        return r;
    }
};

// contexts alive in this process: a context holds pointers into its communicator (transport, peer-to-peer state),
// so replacing the communicator (bicg_comm_init_*, bicg_comm_finalize) orphans them -- they can still be
// destroyed, nothing else
extern std::vector<bicg_ctx *> g_live;

// host vectors of a rank without rows (bicg_ctx::phantom): `count` zeros to read / a place to write
inline const double *host_in(bicg_ctx *c, const double *p, size_t count = 1)
{
    if (!c->phantom) return p;
    c->ph_scratch.assign(std::max<size_t>(count, 1), 0.0);
    return c->ph_scratch.data();
}
inline double *host_out(bicg_ctx *c, double *p, size_t count = 1)
{
    if (!c->phantom || !p) return p;
    if (c->ph_scratch.size() < count) c->ph_scratch.assign(count, 0.0);
    return c->ph_scratch.data();
}

inline void use_device(const bicg_ctx *c)
{
    if (!c->comm)
        die("bicg_ctx", "the communicator this context was built on has been replaced or finalized; only bicg_destroy is valid now");
    BICG_HIP(hipSetDevice(c->device));
}

// ---- section timing (bicg_solver.cpp)
enum { SEC_VEC = 0, SEC_SPMV = 1, SEC_SHIFT = 2, SEC_REDUCE = 3, SEC_COUNT = 4, SEC_STOP = 255 };
constexpr int kMaxSectionMarks = 1 << 16;
void sec_mark(bicg_ctx *c, int label);
void sec_remark(bicg_ctx *c);
void sec_begin(bicg_ctx *c, bool on);
void sec_collect(bicg_ctx *c, int iters);
struct SubSection {    // the enclosed launches are part `sub` of the current product
    bicg_ctx *c; int prev;
    SubSection(bicg_ctx *ctx, int sub) : c(ctx), prev(ctx->cur_sub) { c->cur_sub = sub; sec_remark(c); }
    ~SubSection() { c->cur_sub = prev; sec_remark(c); }
};
struct Section {       // the enclosed launches belong to `label`; afterwards the enclosing section continues
    bicg_ctx *c; int prev;
    Section(bicg_ctx *ctx, int label) : c(ctx), prev(ctx->sec_cur) { if (prev != SEC_STOP) sec_mark(c, label); }
    ~Section() { if (prev != SEC_STOP) sec_mark(c, prev); }
};

// ---- dot groups, products, iterations (bicg_solver.cpp)
Reduce grp_produce(bicg_ctx *c, int off, int n, int phase, unsigned nwg = 0);
Finish grp_desc(bicg_ctx *c, int roles);
void grp_close(bicg_ctx *c, bool local_only = false);
Launch grp_consume(bicg_ctx *c);
Finish grp_for_spmv(bicg_ctx *c);
void group_enqueue(bicg_ctx *c, int n, int phase, hipEvent_t after);
void group_now(bicg_ctx *c, int n, int phase);
void group_defer(bicg_ctx *c, int n, int phase);
void group_flush(bicg_ctx *c);
bool stencil_product(const bicg_ctx *c);
bool hosted(const bicg_ctx *c);      // several ranks whose collectives the host enqueues (RCCL / host transports)
void spmv(bicg_ctx *c, double *xin, double *yout, int ndot, const double *u, Reduce red, Finish fin = Finish{}, int epi = 0,
          Scal *S = nullptr);
void spmv_grp(bicg_ctx *c, double *xin, double *yout, int ndot = 0, const double *u = nullptr, int phase = PH_NONE);
void spmv_epi(bicg_ctx *c, double *xin, double *yout, int epi, int nd, int phase);
void halo_only(bicg_ctx *c, double *xin);
void spmm_pass(bicg_ctx *c, int nvec, const double *sigma_host, bool with_b, bool sigma_staged = false, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void spmm_stage_sigma(bicg_ctx *c, int nvec, const double *sigma_host);
bool spmm_possible(const bicg_ctx *c);
void spmm_buffers(bicg_ctx *c);
void scal_reset(bicg_ctx *c);
void fetch_scal(bicg_ctx *c);
void run_begin(bicg_ctx *c, int method, const bicg_options *opt_in);
int run_iterate(bicg_ctx *c, int nsteps);
int run_end(bicg_ctx *c, bicg_result *res);
int run_solver(bicg_ctx *c, int method, const bicg_options *opt_in, bicg_result *res);
// ---- the shifted family (bicg_shifted.cpp)
int run_shifted(bicg_ctx *c, int mode, double *x_set_host, double *r_host, const double *sigma, int nsig, int seed,
                const bicg_options *opt_in, bicg_result *res);
// ---- plan and context (bicg_create.cpp)
bool all_ranks(Comm *comm, bool mine);
void sell_order_for_big_grids(bicg_ctx *c, uint32_t ngroups);
bool persist_chunk(bicg_ctx *c, int niter);
bool persist_chunk_shifted(bicg_ctx *c, int mode, int niter, int it0, int nsig, int seed, double shift);
void persist_account(bicg_ctx *c);
// ---- drop-in entry points (bicg_dropin.cpp)
void check_square(const INFO_Matrix *info);
void env_options(bicg_options *o);

