// bicg_solver.cpp -- GPU-resident iteration drivers behind include/bicgstab_hip.h (the C ABI itself: bicg_api.cpp, bicg_dropin.cpp;
// contexts and plans: bicg_create.cpp; the shifted family: bicg_shifted.cpp; what they share: bicg_host.h).
//
// One context per rank: the rank's diag/offd CSR blocks, the SpMV plan (row blocks, interior /
// boundary split, halo lists), twelve vectors of rows+halo doubles, and a small device-resident
// scalar block. An iteration is a fixed sequence of kernel launches (and, across ranks, halo
// exchanges and packed all-reduces) with NO host synchronisation: alpha/beta/omega live on the
// device, the convergence test of the reference's while loop (src/solver.c:86) is evaluated on the
// device and turns every later kernel into a no-op, and the host only looks every `check_every`
// iterations.
#include "bicg_host.h"


// ---------------------------------------------------------------- dot groups: consumer-side finish
// (the four solvers of reference src/solver.c; struct Finish in bicg_device.h). A group is PRODUCED by
// one or two kernels (per-wavefront partials), then CONSUMED by the kernel that needs the scalars,
// by an SpMV that only has to deposit the sums, or by the stand-alone finisher.
// ---------------------------------------------------------------- section timing
void sec_mark(bicg_ctx *c, int label)
{
    if (!c->time_sections || label == c->sec_cur) return;
    if (c->sec_used + 1 >= (int)c->sec_ev.size()) {      // pool used up: close the open section, stop marking
        if (c->sec_cur != SEC_STOP && c->sec_used < (int)c->sec_ev.size()) {
            BICG_HIP(hipEventRecord(c->sec_ev[c->sec_used], c->sc));
            c->sec_lab[c->sec_used++] = SEC_STOP;
        }
        c->sec_cur = SEC_STOP; c->time_sections = false; c->sec_exhausted = true;
        return;
    }
    BICG_HIP(hipEventRecord(c->sec_ev[c->sec_used], c->sc));
    c->sec_k[c->sec_used] = c->cur_k; c->sec_sub[c->sec_used] = (unsigned char)((c->cur_prod << 4) | c->cur_sub);
    c->sec_lab[c->sec_used++] = (unsigned char)label;
    c->sec_cur = label;
}
// a mark although the label stays: a new iteration, or another part of the same product
void sec_remark(bicg_ctx *c)
{
    if (!c->time_sections || c->sec_cur == SEC_STOP) return;
    const int label = c->sec_cur;
    c->sec_cur = -1;
    sec_mark(c, label);
}
void sec_begin(bicg_ctx *c, bool on)
{
    c->time_sections = on; c->sec_exhausted = false;
    c->sec_used = 0; c->sec_cur = SEC_STOP; c->sec_iters = 0;
    c->cur_k = 0; c->cur_prod = 0; c->cur_sub = 0; c->switch_sec = 0.0;
    for (double &m : c->sec_ms) m = 0.0;
    if (on && c->sec_ev.empty()) {
        c->sec_ev.resize(kMaxSectionMarks);
        c->sec_lab.resize(kMaxSectionMarks);
        c->sec_k.resize(kMaxSectionMarks); c->sec_sub.resize(kMaxSectionMarks);
        for (auto &e : c->sec_ev) BICG_HIP(hipEventCreate(&e));
    }
}
// after the stream has been synchronised: sum the spans (sections marked so far), covering `iters` iterations
void sec_collect(bicg_ctx *c, int iters)
{
    if (c->sec_used == 0) return;
    for (double &m : c->sec_ms) m = 0.0;
    for (int i = 0; i + 1 < c->sec_used; ++i) {
        if (c->sec_lab[i] == SEC_STOP) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->sec_ev[i], c->sec_ev[i + 1]) == hipSuccess) c->sec_ms[c->sec_lab[i]] += ms;
    }
    c->sec_iters = iters;
}

Reduce grp_produce(bicg_ctx *c, int off, int n, int phase, unsigned nwg)
{
    if (c->grp.active) die("internal", "a dot group was produced while the previous one was still open");
    bicg_ctx::Group &g = c->grp;
    g = bicg_ctx::Group{};
    g.active = true;
    g.seq = ++c->grp_seq;
    g.n = n; g.off = off; g.phase = phase; g.buf = (int)(g.seq & 1u);
    g.nparts = nwg * (kBlock / 64);               // SpMV producers: set by spmv()
    if (c->p2p) g.mail_seq = c->p2p->red_seq++;
    Reduce r{};
    r.partial = c->wpart[g.buf];
    r.wave = 1; r.red_off = off; r.phase = phase;
    return r;
}

Finish grp_desc(bicg_ctx *c, int roles)
{
    const bicg_ctx::Group &g = c->grp;
    Finish f{};
    f.partial = c->wpart[g.buf]; f.nparts = g.nparts; f.seq = g.seq;
    f.shard = c->shard_ll + (size_t)g.buf * kShardLL * kRedSlots * 2;
    f.shard_clear = c->shard_ll + (size_t)(g.buf ^ 1) * kShardLL * kRedSlots * 2;
    f.n = g.n; f.red_off = g.off; f.phase = g.phase; f.roles = roles;
    f.spin_ticks = c->spin_ticks;
    if (c->p2p) { f.p2p = c->p2p->red_desc(g.mail_seq); f.alarm = c->alarm; }
    return f;
}

// stand-alone finisher, in place on the current scalar block. local_only: deposit this rank's sums
// and leave the recurrence to the all-reduce + apply kernel the host enqueues next.
void grp_close(bicg_ctx *c, bool local_only)
{
    if (!c->grp.active) return;
    Finish f = grp_desc(c, FIN_BLOCK0 | (c->grp.staged ? 0 : FIN_SHARDS | FIN_PUSH) | (local_only ? FIN_LOCAL : 0));
    if (local_only) f.phase = PH_NONE;
    launch_finish(Launch{c->S, f, c->sc});
    c->grp.active = false;
    if (c->p2p) c->halo_unsynced = 0;
}

// transports whose collectives the host enqueues (RCCL, host callbacks)
bool hosted(const bicg_ctx *c) { return !c->single() && !c->p2p; }

// Launch descriptor for an element-wise kernel of the four solvers; the open group (if any) is
// finished by that kernel, and everything enqueued afterwards reads the scalar block it writes.
Launch grp_consume(bicg_ctx *c)
{
    Launch L{c->S, Finish{}, c->sc};
    if (!c->grp.active) return L;
    L.fin = grp_desc(c, FIN_APPLY | (c->grp.staged ? 0 : FIN_SHARDS | FIN_PUSH));
    c->cur ^= 1;
    c->S = c->Sbuf + c->cur;
    L.fin.Snext = c->S;
    c->grp.active = false;
    if (c->p2p) c->halo_unsynced = 0;      // an all-reduce is a barrier among the ranks
    return L;
}

// The open group as seen by the next SpMV launch: a deferred group is staged (shards summed, sums on
// their way to the peers) and stays open; anything else is closed first.
Finish grp_for_spmv(bicg_ctx *c)
{
    bicg_ctx::Group &g = c->grp;
    if (!g.active) return Finish{};
    if (hosted(c)) { grp_close(c, true); return Finish{}; }
    if (g.deferred) {
        if (g.staged) return Finish{};
        g.staged = true;
        return grp_desc(c, FIN_SHARDS | FIN_PUSH);
    }
    grp_close(c);
    return Finish{};
}

// ---------------------------------------------------------------- dot groups across ranks
void group_enqueue(bicg_ctx *c, int n, int phase, hipEvent_t after)
{
    if (c->comm->stream_ordered()) {
        BICG_HIP(hipStreamWaitEvent(c->sm, after, 0));
        c->comm->allreduce_sum(c->S->red + c->pend_off, n, c->sm);
        launch_apply(c->S, phase, c->sm);
        hipEvent_t e = c->ev_red[c->i_red++ % kEvRing];
        BICG_HIP(hipEventRecord(e, c->sm));
        c->pend_ev = e;
    } else {
        c->comm->allreduce_sum(c->S->red + c->pend_off, n, c->sc);   // synchronises sc
        launch_apply(c->S, phase, c->sc);
        c->pend_ev = nullptr;
    }
}

// all-reduce the n sums in Scal::red and apply `phase` before anything else runs on the compute
// stream: nothing can overlap, so both are enqueued on the compute stream itself (a round trip
// through the communication stream costs two cross-stream event hand-offs, ~10 us eager)
void group_now(bicg_ctx *c, int n, int phase)
{
    if (c->wave_mode) {
        // single rank / peer-to-peer: the group stays open for the kernel that consumes it
        if (!hosted(c)) { c->grp.deferred = false; return; }
        Section sec(c, SEC_REDUCE);
        const bicg_ctx::Group g = c->grp;
        grp_close(c, true);                                   // this rank's sums -> Scal::red
        c->comm->allreduce_sum(c->S->red + g.off, g.n, c->sc);
        launch_apply(c->S, g.phase, c->sc);
        return;
    }
    if (c->single()) return;   // applied in-kernel by the finishing workgroup
    Section sec(c, SEC_REDUCE);
    if (c->p2p) {              // the producers stored their sums into every rank's mailbox already
        if (c->open_inline) {  // ... and the last of them collects and applies (Reduce::p2p.n_collect)
            c->open_inline = false;
            c->p2p->red_seq++;
        } else {
            launch_apply_p2p(c->S, phase, n, c->p2p->red_desc(c->p2p->red_seq++), c->p2p->timeout_ticks, c->sc);
        }
        c->halo_unsynced = 0;
        return;
    }
    c->pend_off = 0;
    c->comm->allreduce_sum(c->S->red, n, c->sc);
    launch_apply(c->S, phase, c->sc);
}

// same, but the all-reduce is started by the NEXT spmv() after its halo exchange is in flight and
// joined after that SpMV: the overlap of reference src/solver.c:363-367 and 377-385
void group_defer(bicg_ctx *c, int n, int phase)
{
    if (c->wave_mode) {
        if (!hosted(c)) { c->grp.deferred = true; return; }   // staged by the next SpMV launch, applied by the consumer
        const bicg_ctx::Group g = c->grp;
        if (!c->overlap || !c->comm->stream_ordered()) { group_now(c, n, phase); return; }
        grp_close(c, true);
        hipEvent_t e = c->ev_dots[c->i_dots++ % kEvRing];
        BICG_HIP(hipEventRecord(e, c->sc));
        c->pend = true; c->pend_n = g.n; c->pend_off = g.off; c->pend_phase = g.phase; c->pend_ev = e;
        return;
    }
    if (c->single()) return;
    if (c->p2p) {   // collected after the next SpMV: the sums cross the links while it runs
        c->pend = true; c->pend_n = n; c->pend_phase = phase; c->pend_seq = c->p2p->red_seq++;
        return;
    }
    if (!c->overlap || !c->comm->stream_ordered()) { group_now(c, n, phase); return; }
    hipEvent_t e = c->ev_dots[c->i_dots++ % kEvRing];
    BICG_HIP(hipEventRecord(e, c->sc));
    c->pend = true; c->pend_n = n; c->pend_off = 0; c->pend_phase = phase; c->pend_ev = e;
}

// every row on the sliced-ELL path, and the plan found a grid's 7-point stencil (build_stencil_plan)
bool stencil_product(const bicg_ctx *c)
{
    // (whatever order the groups are listed in; across ranks: the halo-free rows are whole planes, StencilDev::z_lo / z_hi)
    return c->st.on && c->nblk == 0 && c->glist_all && (c->single() ? c->ng_bnd == 0 : c->st_multi);
}

// ---------------------------------------------------------------- distributed SpMV
// y = A x (+ fused dots). Replaces MPI_csr_spmv_ovlap (reference src/matrix.c:428-441): the halo
// exchange runs on the communication stream while the interior row blocks are multiplied; row
// blocks that touch the halo run after it has landed. Every row is produced by exactly one
// workgroup as (0 + sum_diag) + sum_offd, the reference's order.
// fin: a dot group of earlier kernels that the first kernel launched here finishes (grp_for_spmv).
void spmv(bicg_ctx *c, double *xin, double *yout, int ndot, const double *u, Reduce red, Finish fin, int epi,
          Scal *S)
{
    Section sec(c, SEC_SPMV);      // halo exchange and the joins of deferred all-reduces included
    SpmvArgs a;
    a.fin = fin;
    a.epi = c->v;
    a.sell = {c->s_val, c->s_col, c->s_base, c->s_len, c->s_col16, c->s_base16, c->sell_jag ? 1 : 0, c->win_ptr, c->win_runs, c->win_slots, c->sell_perm};
    a.sell.ubase = c->s_ubase; a.sell.uoff = c->s_uoff; a.sell.vbase = c->s_vbase; a.sell.uval = c->s_uval; a.sell.mbase = c->s_mbase; a.sell.rmask = c->s_rmask;
    a.sell.sdesc = c->s_desc; a.sell.all_lists = c->sell_all_lists ? 1 : 0; a.sell.uoff8 = c->s_uoff8; a.sell.ystride = c->sell_ystride;
    a.sell.st = c->st;
    a.sell.lane_info = c->jagw_fast ? c->lane_info : nullptr; a.sell.win_max_runs = c->win_max_runs;
    a.sell.win_list = c->win_list; a.sell.win_lptr = c->win_lptr; a.sell.win_ltotal = c->win_ltotal;
    a.glist = nullptr;
    a.nrows = c->n_loc;
    a.diag = {c->d_val, c->d_col, c->d_ptr};
    a.diag_col16 = c->d_col16; a.rowsplit = c->rowsplit ? 1 : 0;
    a.offd = {c->o_val, c->o_col, c->o_ptr};
    a.desc = nullptr; a.nlist = 0;
    a.x = xin; a.y = yout; a.u = u; a.S = S ? S : c->S;
    Scal *const Sh = a.S;        // every kernel of this call reads the same scalar block
    a.nt = c->sell_nt ? 1 : 0;
    a.shift = c->cur_shift; a.has_shift = c->cur_has_shift ? 1 : 0;
    // Up to four launches share one dot group (one partial slot per workgroup, numbered in launch
    // order): {sliced-ELL groups, CSR row blocks} x {interior, halo-touching}.
    a.groups_per_wg = ndot > 0 ? c->sell_gpw_dots : c->sell_gpw;
    // Ranks SHARING a GPU (tests; bicg_ctx::wg_cap): a launch with the halo exchange inside may consist of workgroups that ALL wait
    // for another rank's values (a numbering without locality: every 256-row group touches the halo). The first rank's launch would
    // fill the device with waiting workgroups and keep the launches that hold the awaited pushes out until the time-out; every
    // rank's launch, pushing workgroups included, has to fit beside the others'.
    if (c->wg_cap && c->p2p && c->ll_fused) {
        const unsigned room = c->wg_cap > 80u ? c->wg_cap - 64u : 16u, ng = c->ng_int + c->ng_bnd;
        a.groups_per_wg = std::max<int>(a.groups_per_wg, (int)((ng + room - 1) / room));
    }
    a.xcd_map = c->sell_xcd;
    // Consecutive products of a solve run over the matrix in alternating directions (BICG_SELL_ALT=0: always forward): matrix +
    // vectors of a Transport-sized system exceed the 256 MiB Infinity Cache by a quarter, so a product that starts where the
    // previous one ended finds the most recently streamed part of the matrix still cached, while cyclic forward passes
    // evict it just before it is needed. Rows, hence results of the product, are unaffected; the dot partials of a
    // reversed launch land in mirrored slots (a different, equally fixed association).
    a.reverse = (c->sell_alt && c->single()) ? (c->spmv_dir ^= 1) : 0;
    // the plane-marching product (bicg_stencil.hip) takes the rank's halo-free planes in one launch of its own tiling; across ranks
    // the halo-touching planes follow behind the exchange through the slice-by-slice kernel, as separate launches (never the
    // launch with the exchange inside)
    const bool stencil = stencil_product(c) && (epi == 0 || epi == 3) && !a.has_shift;
    if (epi == 3 && !(stencil && c->single())) die("internal", "CA-BiCGStab's fused q / y epilogue without the plane-marching product");
    const unsigned g_si = stencil ? stencil_grid(c->st) : sell_grid(c->ng_int, a.groups_per_wg), g_ci = spmv_grid(c->n_int);
    const unsigned g_sb = sell_grid(c->ng_bnd, a.groups_per_wg), g_cb = spmv_grid(c->n_bnd);
    const bool fused = c->p2p && c->ll_fused && !stencil;
    const bool merged = !c->single() && !stencil && (fused || (!c->p2p && !(c->comm->stream_ordered() && c->overlap) && c->glist_all));
    const unsigned g_sall = sell_grid(c->ng_int + c->ng_bnd, a.groups_per_wg);
    red.expected = merged ? g_sall + g_ci + g_cb : g_si + g_ci + g_sb + g_cb;
    red.slot_base = 0;
    a.red = red;
    if (red.wave && (ndot > 0 || epi)) c->grp.nparts = red.expected * (kBlock / 64);   // one partial per wavefront

    // per-kernel timing: every SpMV kernel of this call gets its own start/stop event pair
    const bool timed = c->time_kernels && c->tev_used + 8 <= (int)c->tev.size();
    bool any_timed = false;
    auto ev = [&](int i) -> hipEvent_t { return timed ? c->tev[c->tev_used + i] : nullptr; };
    auto took = [&](bool launched) {
        if (launched) a.fin.seq = 0;          // the first kernel of this SpMV finished the open group
        if (launched && timed) { c->tev_used += 2; any_timed = true; }
    };

    auto interior = [&]() {
        a.glist = c->glist_int_identity ? nullptr : c->glist_int; a.nlist = c->ng_int; a.red.slot_base = 0;
        if (stencil) {
            // (the plan only selects tilings the kernel is built for, and the epilogue only with ticket reductions: a launch that
            // is declined here would leave y unwritten and a ticket group waiting for partial sums that never come)
            if (!launch_spmv_stencil(a, ndot, epi == 3 ? 1 : 0, c->sc, ev(0), ev(1)))
                die("internal", "the plane-marching product declined a launch its plan had selected (lines per wavefront / reduction mode)");
            took(true);
        } else {
            took(launch_spmv_sell(a, ndot, false, c->sc, ev(0), ev(1)));
        }
        a.desc = c->desc_int; a.nlist = c->n_int; a.red.slot_base = g_si;
        took(launch_spmv(a, ndot, false, c->sc, ev(0), ev(1)));
    };
    auto boundary = [&]() {
        a.glist = c->glist_bnd; a.nlist = c->ng_bnd; a.red.slot_base = g_si + g_ci;
        took(launch_spmv_sell(a, ndot, true, c->sc, ev(0), ev(1)));
        a.desc = c->desc_bnd; a.nlist = c->n_bnd; a.red.slot_base = g_si + g_ci + g_sb;
        took(launch_spmv(a, ndot, true, c->sc, ev(0), ev(1)));
    };

    if (epi && !(c->glist_all && c->nblk == 0 && (c->single() || fused))) die("internal", "SpMV epilogue on a multi-launch SpMV");
    // Epilogue launches publish one partial row per WAVEFRONT for 32 helper workgroups to add: beyond a few thousand
    // workgroups that sum, not the matrix, sets the pace (16.8 M rows = 65 k workgroups: 4.6 instead of 1.25 ms per
    // iteration), so a workgroup takes several 256-row groups. Ranks sharing a GPU (tests): every row workgroup of
    // the launch waits for the scalars, all ranks' launches must fit on the GPU together.
    const unsigned epi_cap = c->wg_cap ? c->wg_cap : 8192u;
    if (epi && !stencil && c->ng_int + c->ng_bnd > epi_cap) {
        const unsigned ng = c->ng_int + c->ng_bnd;
        a.groups_per_wg = std::max<int>(a.groups_per_wg, (int)((ng + epi_cap - 1) / epi_cap));
        red.expected = sell_grid(ng, a.groups_per_wg) + g_ci;
        a.red.expected = red.expected;
        c->grp.nparts = red.expected * (kBlock / 64);
    }
    if (c->single()) {
        if (epi && !stencil) {
            a.glist = nullptr; a.nlist = c->ng_int; a.red.slot_base = 0;
            took(launch_spmv_sell_epi(a, epi, false, c->sc, ev(0), ev(1)));
        } else {
            interior();
        }
    } else if (c->p2p) {
        // peer-to-peer: the send list is stored straight into the landing rings of the ranks that
        // need it, the interior rows run while the values cross the links, one kernel decodes the
        // ring slot into the halo tail of x, then the rows that touch the halo run
        if (c->halo_unsynced >= kHaloRing - 2) {   // nothing has throttled the senders for a while
            launch_p2p_barrier(c->p2p->red_desc(c->p2p->bar_seq++), c->p2p->timeout_ticks, Sh, c->sc);
            c->halo_unsynced = 0;
        }
        const unsigned seq = ++c->halo_seq;
        c->halo_unsynced++;
        const bool lose = c->fault_after > 0 && seq >= (unsigned)c->fault_after;   // BICG_TEST="p2p-fault-after=n" (tests)
        if (fused) {
            // ONE launch: leading workgroups push, the others multiply; halo-touching groups come last
            // and read the landing ring directly
            a.ll.ring = c->halo_ring; a.ll.halo = c->halo; a.ll.seq = seq;
            a.ll.nsend = lose ? 0u : c->nsend;
            a.ll.npush = c->nsend ? std::min<unsigned>((c->nsend + kBlock - 1) / kBlock, 64u) : 0u;
            a.ll.first_bnd = c->ng_int;
            a.ll.send_idx = c->send_idx; a.ll.dst0 = c->push_dst0; a.ll.dstride = c->push_stride;
            a.ll.timeout_ticks = c->p2p->timeout_ticks;
            a.glist = c->glist_ll; a.nlist = c->ng_int + c->ng_bnd; a.red.slot_base = 0;
            if (epi) took(launch_spmv_sell_epi(a, epi, true, c->sc, ev(0), ev(1), true));
            else took(launch_spmv_sell(a, ndot, true, c->sc, ev(0), ev(1), true));
            a.glist = nullptr;
            a.desc = c->desc_int; a.nlist = c->n_int; a.red.slot_base = g_sall;
            took(launch_spmv(a, ndot, false, c->sc, ev(0), ev(1)));
        } else {
            if (!lose) launch_halo_push(xin, c->send_idx, c->nsend, c->push_dst0, c->push_stride, seq, Sh, c->sc);
            interior();
            launch_halo_unpack(c->halo_ring, c->halo, seq, xin + c->n_loc, Sh, c->p2p->timeout_ticks, c->sc);
            boundary();
        }
        if (c->pend) {
            c->pend = false;
            launch_apply_p2p(c->S, c->pend_phase, c->pend_n, c->p2p->red_desc(c->pend_seq), c->p2p->timeout_ticks, c->sc);
            c->halo_unsynced = 0;
        }
    } else {
        const bool two_streams = c->comm->stream_ordered() && c->overlap;
        hipEvent_t eh = nullptr;
        c->cur_sub = 1; sec_remark(c);          // the exchange: the reference's "agv" section (src/matrix.c:432)
        launch_halo_pack(xin, c->send_idx, c->nsend, c->sendbuf, Sh, c->sc);
        if (two_streams) {
            hipEvent_t ep = c->ev_pack[c->i_pack++ % kEvRing];
            BICG_HIP(hipEventRecord(ep, c->sc));
            BICG_HIP(hipStreamWaitEvent(c->sm, ep, 0));
            c->comm->exchange(c->sendbuf, c->scnt.data(), c->sdsp.data(), xin + c->n_loc, c->rcnt.data(), c->rdsp.data(), c->sm);
            eh = c->ev_halo[c->i_halo++ % kEvRing];
            BICG_HIP(hipEventRecord(eh, c->sm));
        } else {
            c->comm->exchange(c->sendbuf, c->scnt.data(), c->sdsp.data(), xin + c->n_loc, c->rcnt.data(), c->rdsp.data(), c->sc);
        }
        c->cur_sub = 0; sec_remark(c);
        bool joined_pending = false;
        if (c->pend) {   // start the deferred all-reduce behind the halo traffic
            hipEvent_t after = c->pend_ev;
            c->pend = false;
            group_enqueue(c, c->pend_n, c->pend_phase, after);
            joined_pending = true;
        }
        if (!two_streams && c->glist_all && !stencil) {
            // nothing overlaps the exchange: one launch over ALL sliced-ELL groups (rows without offd
            // entries simply find an empty offd range) instead of an interior + a boundary launch
            a.glist = nullptr; a.nlist = c->ng_int + c->ng_bnd; a.red.slot_base = 0;
            took(launch_spmv_sell(a, ndot, true, c->sc, ev(0), ev(1)));
            a.desc = c->desc_int; a.nlist = c->n_int; a.red.slot_base = g_sall;
            took(launch_spmv(a, ndot, false, c->sc, ev(0), ev(1)));
            a.desc = c->desc_bnd; a.nlist = c->n_bnd; a.red.slot_base = g_sall + g_ci;
            took(launch_spmv(a, ndot, true, c->sc, ev(0), ev(1)));
        } else {
            interior();
            if (eh) BICG_HIP(hipStreamWaitEvent(c->sc, eh, 0));
            c->cur_sub = 2; sec_remark(c);      // the rows that touch the halo: the reference's second mult() (src/matrix.c:440)
            boundary();
            c->cur_sub = 0; sec_remark(c);
        }
        if (joined_pending && c->pend_ev) {
            BICG_HIP(hipStreamWaitEvent(c->sc, c->pend_ev, 0));
            c->pend_ev = nullptr;
        }
    }
    if (a.fin.seq) {   // no SpMV kernel was launched (a rank without work): finish the group on its own
        Finish f = a.fin;
        f.roles |= FIN_BLOCK0;
        launch_finish(Launch{c->S, f, c->sc});
    }
    if (any_timed) c->spmv_calls_timed++;
}


// SpMV of the pipelined solvers (consumer-side finish): a deferred group of earlier kernels is staged
// by this launch; the SpMV's own dots (ndot > 0) open the next group.
void spmv_grp(bicg_ctx *c, double *xin, double *yout, int ndot, const double *u, int phase)
{
    const Finish fin = grp_for_spmv(c);
    Reduce red{};
    if (ndot > 0) {
        if (c->grp.active) {     // a deferred group is still open: an SpMV with dots of its own cannot carry it
            if (fin.seq) die("internal", "an SpMV with dots was asked to stage a deferred group");
            grp_close(c);
        }
        red = grp_produce(c, 0, ndot, phase, 0);
    }
    spmv(c, xin, yout, ndot, u, red, fin);
}

// The halo exchange of spmv() on its own: afterwards xin[rows .. rows + halo) holds the other ranks' values.
void halo_only(bicg_ctx *c, double *xin)
{
    if (c->single() || (c->halo == 0 && c->nsend == 0)) return;
    if (c->p2p) {
        if (c->halo_unsynced >= kHaloRing - 2) {
            launch_p2p_barrier(c->p2p->red_desc(c->p2p->bar_seq++), c->p2p->timeout_ticks, c->S, c->sc);
            c->halo_unsynced = 0;
        }
        const unsigned seq = ++c->halo_seq;
        c->halo_unsynced++;
        launch_halo_push(xin, c->send_idx, c->nsend, c->push_dst0, c->push_stride, seq, c->S, c->sc);
        launch_halo_unpack(c->halo_ring, c->halo, seq, xin + c->n_loc, c->S, c->p2p->timeout_ticks, c->sc);
        return;
    }
    launch_halo_pack(xin, c->send_idx, c->nsend, c->sendbuf, c->S, c->sc);
    c->comm->exchange(c->sendbuf, c->scnt.data(), c->sdsp.data(), xin + c->n_loc, c->rcnt.data(), c->rdsp.data(), c->sc);
}

// Y_j = (A + sigma_j I) X_j for nvec <= kSpmmCols vectors that sit shift-major in c->mm_in: one pass over A.
// with_b: c->v.b holds b, c->mm_out receives || b - Y_j ||^2 (this rank's rows); otherwise c->mm_yt receives Y.
// the shifts of a pass go to the device BEFORE the pass is timed: the upload ends in a host-side wait (the values live on the
// caller's stack frame), which used to sit inside bicg_spmm's event bracket -- 30-40 us of an idle GPU counted as kernel time
void spmm_stage_sigma(bicg_ctx *c, int nvec, const double *sigma_host)
{
    double sg[kSpmmCols] = {0};
    for (int j = 0; j < nvec; ++j) sg[j] = sigma_host[j];
    BICG_HIP(hipMemcpyAsync(c->mm_sigma, sg, sizeof sg, hipMemcpyHostToDevice, c->sc));
    BICG_HIP(hipStreamSynchronize(c->sc));     // sg lives on this stack frame
}

// e0 / e1 (optional): stamped at the start / end of the SpMM KERNEL where the launch can carry them (the windowed and the pipelined
// form), else around the pass
void spmm_pass(bicg_ctx *c, int nvec, const double *sigma_host, bool with_b, bool sigma_staged, hipEvent_t e0, hipEvent_t e1)
{
    const size_t st = c->stride;
    for (int j = 0; j < nvec; ++j) halo_only(c, c->mm_in + (size_t)j * st);
    // the windowed form (k_spmm_win) reads the shift-major vectors directly and writes Y shift-major into mm_yt
    const unsigned wslots = c->win_slots ? c->win_slots : (c->s_col16 && !c->sell_jag && c->fw.ncl > 0 ? c->fw.slots : 0u);
    c->mm_win = c->mm_win_env != 0 && spmm_win_vectors(wslots) > 0;
    if (!c->mm_win) launch_rows_from_vectors(c->mm_in, st, nvec, c->n_loc + c->halo, c->mm_xt, c->sc);
    SpmmArgs a{};
    a.sell = {c->s_val, c->s_col, c->s_base, c->s_len, c->s_col16, c->s_base16, c->sell_jag ? 1 : 0, c->win_ptr, c->win_runs, c->win_slots, c->sell_perm};
    a.sell.win_list = c->win_list; a.sell.win_lptr = c->win_lptr; a.sell.win_ltotal = c->win_ltotal;
    a.sell.lane_info = c->lane_info; a.sell.win_max_runs = c->win_max_runs;
    a.dptr = c->d_ptr; a.offd = {c->o_val, c->o_col, c->o_ptr};
    a.nrows = c->n_loc; a.ngroups = c->ng_int + c->ng_bnd;
    a.xt = c->mm_xt; a.yt = with_b ? nullptr : c->mm_yt; a.b = with_b ? c->v.b : nullptr; a.partial = c->mm_part;
    a.xcd_map = c->mm_xcd ? 1 : 0;
    if (sigma_host) {
        if (!sigma_staged) spmm_stage_sigma(c, nvec, sigma_host);
        a.sigma = c->mm_sigma;
    }
    c->mm_dma = false;
    if (c->mm_win) {
        a.xs = c->mm_in; a.ys = with_b ? nullptr : c->mm_yt; a.vstride = st; a.nvec = nvec; a.wslots = wslots;
        a.tail_most = c->jag_tail16_max;
        if (const char *sv = test_tok("spmm-skip")) a.dbg = atoi(sv);
        if (!c->win_slots) a.cl = c->fw;
        // the pipelined form where the block qualifies (BICG_PLAN="spmm-window=1": k_spmm_win everywhere)
        c->mm_dma = c->mm_win_env == 3 && !a.dbg && launch_spmm_pipe(a, !c->single(), c->sc, e0, e1) == hipSuccess;
        // ... and its form for ragged rows (jagged slices with x windows: bicg_spmm_jag.hip)
        if (!c->mm_dma && c->mm_win_env == 3 && c->win_slots && c->win_near16) c->mm_dma = launch_spmm_jpipe(a, !c->single(), c->sc, e0, e1) == hipSuccess;
        if (!c->mm_dma && launch_spmm_win(a, !c->single(), c->sc, e0, e1) != hipSuccess) die("bicg_spmm", "the windowed kernel could not be launched (BICG_PLAN=spmm-window=0 selects the row-major form)");
    } else {
        if (e0) BICG_HIP(hipEventRecord(e0, c->sc));
        launch_spmm_sell(a, !c->single(), c->sc);
        if (e1) BICG_HIP(hipEventRecord(e1, c->sc));
    }
    if (with_b) launch_colsum(c->mm_part, spmm_grid(a.ngroups, a.xcd_map != 0), c->mm_out, c->sc);
}

// every row on the sliced-ELL path, and 32-bit byte offsets into the row-major X (128 B per row) suffice
bool spmm_possible(const bicg_ctx *c)
{
    // (x windows: the kernel keeps a group's runs in 64 LDS entries -- or reads the group's column list, round 6)
    return c->glist_all && c->nblk == 0 && c->sell_entries > 0 && (c->win_max_runs <= 64 || c->win_list) && (uint64_t)c->stride < (1ull << 25);
}

void spmm_buffers(bicg_ctx *c)
{
    if (c->mm_in) return;
    const size_t st = c->stride, ngroups = c->ng_int + c->ng_bnd;
    c->mm_in = dev_alloc<double>((size_t)kSpmmCols * st + 64);      // (+64: k_spmm_pipe copies 16-byte pairs, the last one may reach one column past a vector)
    c->mm_xt = dev_alloc<double>((size_t)kSpmmCols * st);
    c->mm_yt = dev_alloc<double>((size_t)kSpmmCols * st);
    c->mm_part = dev_alloc<double>((ngroups + 8) * kSpmmCols);
    c->mm_out = dev_alloc<double>(kSpmmCols);
    c->mm_sigma = dev_alloc<double>(kSpmmCols);
    BICG_HIP(hipMemset(c->mm_in, 0, sizeof(double) * kSpmmCols * st));
    BICG_HIP(hipDeviceSynchronize());      // the memset ran on the null stream: c->sc does not wait for it
    c->mm_xcd = !(knob_x("BICG_SPMM_XCD") && atoi(knob_x("BICG_SPMM_XCD")) == 0);
    c->mm_win_env = plan_tok("spmm-window") ? atoi(plan_tok("spmm-window")) : 3;      // 3: pipelined form where possible, else windowed
}

// SpMV whose epilogue runs a pipelined phase on the workgroup's own rows (k_spmv_sell_epi): the open dot group is
// summed by the launch's first workgroups and applied at the epilogue; the phase's own nd dots open the next group.
void spmv_epi(bicg_ctx *c, double *xin, double *yout, int epi, int nd, int phase)
{
    const Launch L = grp_consume(c);
    Reduce red = grp_produce(c, 0, nd, phase, 0);
    spmv(c, xin, yout, 0, nullptr, red, L.fin, epi, L.S);
}

// a deferred group that no SpMV picked up (defensive)
void group_flush(bicg_ctx *c)
{
    if (c->wave_mode && !hosted(c)) return;      // consumed by the next element-wise kernel or by fetch_scal
    if (!c->pend) return;
    Section sec(c, SEC_REDUCE);
    if (c->p2p) {
        c->pend = false;
        launch_apply_p2p(c->S, c->pend_phase, c->pend_n, c->p2p->red_desc(c->pend_seq), c->p2p->timeout_ticks, c->sc);
        c->halo_unsynced = 0;
        return;
    }
    hipEvent_t after = c->pend_ev;
    c->pend = false;
    group_enqueue(c, c->pend_n, c->pend_phase, after);
    if (c->pend_ev) BICG_HIP(hipStreamWaitEvent(c->sc, c->pend_ev, 0));
    c->pend_ev = nullptr;
}

void fetch_scal(bicg_ctx *c);


// ---------------------------------------------------------------- the four iterations
struct Driver {
    bicg_ctx *c;
    int method;
    int krr, nrr;
    Vecs &v;
    unsigned vg;       // workgroups of an element-wise kernel

    Driver(bicg_ctx *ctx, int m, int kr, int nr) : c(ctx), method(m), krr(kr), nrr(nr), v(ctx->v), vg(vec_grid(ctx->v.n)) {}

    // element-wise kernel without / with dots: it finishes the open group, then opens its own
    template <class Fn> void vec(Fn launch)
    {
        const Launch L = grp_consume(c);
        launch(v, L);
    }
    template <class Fn> void vec_dots(Fn launch, int n, int phase)
    {
        const Launch L = grp_consume(c);
        const Reduce r = grp_produce(c, 0, n, phase, vg);
        launch(v, L, r);
    }

    // plain and CA-BiCGStab: ticket reductions, scalars applied in place by the producer's last workgroup
    Launch here() const { return Launch{c->S, Finish{}, c->sc}; }

    void init()
    {
        const bool plain = method == BICG_BICGSTAB;
        const bool rr = method == BICG_PIPE_BICGSTAB_RR || (method == BICG_PIPE_BICGSTAB && c->opt.rr_drift > 0.0);
        if (!c->wave_mode) {
            spmv(c, v.x, v.ax, 0, nullptr, c->red(0, PH_NONE));                       // Ax = A x0
            launch_init_residual(v, plain, rr, here(), c->red(0, PH_INIT, true, 1));  // r = b - Ax, r# = r, (r,r)
            group_now(c, 1, PH_INIT);
            if (plain) return;
            spmv(c, v.r, v.w, 1, v.r, c->red(0, PH_INIT_ALPHA, true, 1));             // w = A r, (r,w)
            group_now(c, 1, PH_INIT_ALPHA);
            return;
        }
        spmv_grp(c, v.x, v.ax);                                                    // Ax = A x0
        vec_dots([&](const Vecs &vv, const Launch &L, Reduce r) { launch_init_residual(vv, plain, rr, L, r); }, 1, PH_INIT);
        group_now(c, 1, PH_INIT);
        spmv_grp(c, v.r, v.w, 1, v.r, PH_INIT_ALPHA);                              // w = A r, (r,w)
        group_defer(c, 1, PH_INIT_ALPHA);                                           // overlaps t = A w (src/solver.c:339-343)
        spmv_grp(c, v.w, v.t);
        group_flush(c);
    }

    void iter_plain()   // reference src/solver.c:88-119
    {
        spmv(c, v.p, v.s, 1, v.rh, c->red(0, PH_PLAIN_ALPHA, true, 1));   // s = A p, (r#,s) -> alpha
        group_now(c, 1, PH_PLAIN_ALPHA);
        launch_plain_q(v, here());                              // q = r - alpha s
        spmv(c, v.r, v.y, 2, v.r, c->red(0, PH_OMEGA, true, 2));          // y = A q, (q,y), (y,y) -> omega
        group_now(c, 2, PH_OMEGA);
        launch_plain_xr(v, here(), c->red(0, PH_PLAIN_END, true, 2));    // x, r, (r,r), (r#,r) -> beta, k++
        group_now(c, 2, PH_PLAIN_END);
        launch_plain_p(v, here());                              // p = r + beta (p - omega s)
    }

    void iter_ca()      // reference src/solver.c:217-251
    {
        launch_ca_ps(v, here());                                // p, s recurrences
        if (c->ca_fuse && c->single() && stencil_product(c) && !c->cur_has_shift) {
            // z = A s with q = r - alpha s, y = w - alpha z, (q,y), (y,y) on the product's own rows: s_i and z_i are registers there
            spmv(c, v.s, v.z, 2, nullptr, c->red(0, PH_OMEGA, true, 2), Finish{}, 3);
        } else {
            spmv(c, v.s, v.z, 0, nullptr, c->red(0, PH_NONE));   // z = A s
            launch_qy(v, here(), c->red(0, PH_OMEGA, true, 2)); // q, y, (q,y), (y,y) -> omega
        }
        group_now(c, 2, PH_OMEGA);
        launch_ca_xr(v, here(), c->red(0, PH_NONE, false));     // x, r, (r,r), (r#,r), (r#,s), (r#,z)
        spmv(c, v.r, v.w, 1, v.rh, c->red(2, PH_RECUR_END, true, 5));     // w = A r, (r#,w) -> beta, alpha, k++
        group_now(c, 5, PH_RECUR_END);
    }

    bool replaces(int it) const { return method == BICG_PIPE_BICGSTAB_RR && krr > 0 && (it % krr == 0) && it > 0 && it <= krr * nrr; }
    // two launches per iteration (phases in the SpMV epilogues): every row on the sliced-ELL path and a single
    // SpMV launch per product (one rank, or the peer-to-peer exchange folded into the launch)
    bool fused() const
    {
        return c->fuse_pipe && c->wave_mode && !hosted(c) && c->fuse_plan_ok;
    }

    // last: the caller looks at x / r after this iteration (end of a run_iterate call, adaptive replacement check):
    // phase 1 of the next iteration, which overwrites r with q, must not have run yet
    void iter_pipe(int it, bool force_replace, bool last)   // reference src/solver.c:352-390 and 494-548
    {
        const bool replace = force_replace || replaces(it);
        if (!replace) {
            if (!c->f1_done) {
                vec_dots(launch_pipe_f1, 2, PH_OMEGA);                   // p, s, z, q, y, (q,y), (y,y)
                group_defer(c, 2, PH_OMEGA);
            }
            c->f1_done = false;
            if (fused()) {
                spmv_epi(c, v.z, v.v, 1, 5, PH_RECUR_END);               // v = A z ; x, r, w, five dots   || all-reduce of (q,y), (y,y)
                group_defer(c, 5, PH_RECUR_END);
                if (!last && !replaces(it + 1)) {
                    spmv_epi(c, v.w, v.t, 2, 2, PH_OMEGA);               // t = A w ; phase 1 of iteration it + 1   || all-reduce of the five
                    group_defer(c, 2, PH_OMEGA);
                    c->f1_done = true;
                } else {
                    spmv_grp(c, v.w, v.t);                               // t = A w   || all-reduce
                }
            } else {
                spmv_grp(c, v.z, v.v);                                   // v = A z   || all-reduce
                vec_dots(launch_pipe_f2, 5, PH_RECUR_END);               // x, r, w, five dots
                group_defer(c, 5, PH_RECUR_END);
                spmv_grp(c, v.w, v.t);                                   // t = A w   || all-reduce
            }
        } else {
            if (c->f1_done) die("internal", "replacement step after phase 1 of the same iteration has run");
            vec(launch_p_update);
            spmv_grp(c, v.p, v.s);                                   // s = A p
            spmv_grp(c, v.s, v.z);                                   // z = A s
            vec_dots(launch_qy, 2, PH_OMEGA);
            group_defer(c, 2, PH_OMEGA);
            spmv_grp(c, v.z, v.v);                                   // v = A z
            vec(launch_x_update);
            spmv_grp(c, v.x, v.ax);                                  // Ax = A x
            vec(launch_true_residual);                               // r = b - Ax
            spmv_grp(c, v.r, v.w);                                   // w = A r
            vec_dots(launch_dots5, 5, PH_RECUR_END);
            group_defer(c, 5, PH_RECUR_END);
            spmv_grp(c, v.w, v.t);                                   // t = A w
        }
        group_flush(c);
    }

    void iterate(int it, bool force_replace = false, bool last = true)
    {
        switch (method) {
        case BICG_BICGSTAB: iter_plain(); break;
        case BICG_CA_BICGSTAB: iter_ca(); break;
        default: iter_pipe(it, force_replace, last); break;
        }
    }

    // adaptive residual replacement (additive): true residual vs recursive one
    double drift()
    {
        spmv_grp(c, v.x, v.ax);
        vec_dots(launch_drift, 2, PH_NONE);
        group_now(c, 2, PH_NONE);
        fetch_scal(c);
        return c->hS->red[1] > 0.0 ? sqrt(c->hS->red[0] / c->hS->red[1]) : 0.0;
    }
};

// a fresh scalar block and ticket counters for a stand-alone kernel (bicg_spmv, bicg_dot, the form probe)
void scal_reset(bicg_ctx *c)
{
    c->wave_mode = false;
    c->grp = bicg_ctx::Group{};
    c->spmv_dir = 0;             // stand-alone products and dots: always the same direction, whatever ran before
    BICG_HIP(hipMemsetAsync(c->S, 0, sizeof(Scal), c->sc));
    BICG_HIP(hipMemsetAsync(c->counter, 0, sizeof(unsigned) * (kShards + 1) * kCounterStride, c->sc));
}
bool all_ranks(Comm *comm, bool mine);

void fetch_scal(bicg_ctx *c)
{
    if (c->wave_mode) grp_close(c);      // the host wants the scalars: finish the open group now
    BICG_HIP(hipMemcpyAsync(c->hS, c->S, sizeof(Scal), hipMemcpyDeviceToHost, c->sc));
    if (c->p2p || c->persist_on) BICG_HIP(hipMemcpyAsync(c->h_alarm, c->alarm, sizeof(int), hipMemcpyDeviceToHost, c->sc));
    BICG_HIP(hipStreamSynchronize(c->sc));
    if (c->sm) BICG_HIP(hipStreamSynchronize(c->sm));
    if (c->hS->comm_error || ((c->p2p || c->persist_on) && *c->h_alarm)) {
        c->hS->comm_error = 1; c->hS->done = 1;
        // BICG_P2P_SOFT_FAIL=1: report through bicg_comm_failed() and stop iterating instead of
        // exiting (bench.py then falls back to the RCCL collectives)
        const char *soft = getenv("BICG_P2P_SOFT_FAIL");
        // one rank, no peer-to-peer path: the only waits are those between the workgroups of a persistent launch (they spin on each
        // other and need to be co-resident: another long-running kernel on the same GPU can starve them) or on a dot group's producers
        if (!c->p2p && !c->soft_fail && (!soft || atoi(soft) == 0))
            die("persistent kernel", "workgroups waited for each other longer than the time-out -- is another kernel holding CUs of this GPU? "
                                     "(BICG_PERSIST=0 selects the multi-launch iteration)");
        if (!c->soft_fail && (!soft || atoi(soft) == 0))
            die("peer-to-peer transport", "timed out waiting for another rank (BICG_P2P_TIMEOUT_MS)");
        if (!c->comm_failed)
            fprintf(stderr, "bicgstab_hip: rank %d: peer-to-peer transport timed out waiting for another rank\n", c->rank);
        c->comm_failed = true;
    }
}

// A solve in three steps so that callers (and bench.py) can time exactly K iterations:
// run_begin = set-up phase of the reference (src/solver.c:74-83 etc.), run_iterate = up to n more
// iterations of its while loop, run_end = summary lines and result.
void run_begin(bicg_ctx *c, int method, const bicg_options *opt_in)
{
    bicg_options &o = c->opt;
    if (opt_in) o = *opt_in; else bicg_default_options(&o);
    if (method < BICG_BICGSTAB || method > BICG_PIPE_BICGSTAB_RR) die("bicg_run", "unknown method");
    if (o.max_iter < 0) o.max_iter = 0;
    if (o.check_every < 1) o.check_every = 1;
    c->method = method;
    use_device(c);
    // Plain and CA-BiCGStab need every scalar right after the kernel that produces its sums: the ticket
    // chain at the end of the producer (memory system draining) is then the shortest path. The
    // pipelined solvers defer their groups across an SpMV (src/solver.c:363-367, 377-385): there the sums
    // are staged by that SpMV and finished by the kernel that consumes them, off the critical path.
    c->wave_mode = method >= BICG_PIPE_BICGSTAB;
    c->grp = bicg_ctx::Group{};
    c->f1_done = false;
    c->spmv_dir = 0;             // every solve starts in the same direction (its first product toggles this to 1 = reversed):
                                 // run-to-run bit reproducibility, also with several groups per workgroup
    // Matrix stream policy. The Infinity Cache (256 MiB) is shared by the matrix stream and the
    // solver's vectors. If matrix + vectors exceed it by less than ~25 % ordinary loads win: a good
    // part of the matrix survives from one SpMV to the next (Transport, plain: 149.5 vs 155.0 us
    // per iteration). Beyond that the matrix only evicts the vectors and is streamed with
    // non-temporal loads instead (Transport, pipelined, 10 vectors: 167.3 vs 173.8 us).
    {
        static const int nvec[4] = {6, 8, 10, 11};
        const double ws = (double)c->matrix_bytes + 8.0 * c->stride * nvec[method];
        // (round 4: with the products alternating direction, ordinary loads pay up to 35 % over the cache -- CA-BiCGStab
        // 166.9 -> 150.6 us per iteration; the pipelined solvers' ten vectors are past that: 157.2 vs 160.9)
        // (round 5, the products of rounds 4-5 and an irregular matrix: ordinary loads win far beyond that -- FEM-like, 1.6 M rows,
        // pipelined, 1.42 x the cache: 161 against 180 us; 2.4 M rows plain 1.85 x: 220 against 239; 3.2 M rows plain 2.47 x: 294
        // against 306, pipelined 2.85 x: 338 against 325 -- the cross-over is near 2.5 x: profiles/r05/ab_matrix_stream_policy.txt)
        c->sell_nt = ws > (c->sell_alt ? 2.5 : 1.25) * 256.0 * 1048576.0;
        if (c->sell_nt_env >= 0) c->sell_nt = c->sell_nt_env != 0;
    }

    // trace storage: the (r,r) history is always kept (progress lines), 4 arrays of max_iter
    if (c->trace_cap < o.max_iter) {
        if (c->trace) BICG_HIP(hipFree(c->trace));
        c->trace_cap = o.max_iter > 0 ? o.max_iter : 1;
        c->trace = dev_alloc<double>(4 * (size_t)c->trace_cap);
    }
    Scal h;
    memset(&h, 0, sizeof h);
    h.tol2 = o.tol * o.tol;
    h.max_iter = o.max_iter;
    h.tr_alpha = c->trace;
    h.tr_omega = c->trace + c->trace_cap;
    h.tr_beta = c->trace + 2 * (size_t)c->trace_cap;
    h.tr_dotr = c->trace + 3 * (size_t)c->trace_cap;
    for (int i = 0; i < 2; ++i)       // both scalar blocks: the idle one must not carry `done` of an earlier solve
        BICG_HIP(hipMemcpyAsync(c->Sbuf + i, &h, sizeof h, hipMemcpyHostToDevice, c->sc));
    BICG_HIP(hipMemsetAsync(c->counter, 0, sizeof(unsigned) * (kShards + 1) * kCounterStride, c->sc));
    BICG_HIP(hipMemsetAsync(c->alarm, 0, sizeof(int), c->sc));
    if (c->waitlog) BICG_HIP(hipMemsetAsync(c->waitlog, 0, sizeof(unsigned) * 3 * kWaitCap, c->sc));
    // every work vector starts at zero: defines the reads of p, s, z, v that the reference makes
    // before writing them (src/solver.c:217-222, 352-360) and keeps halo tails finite
    const size_t st = c->stride;
    BICG_HIP(hipMemsetAsync(c->slab + 2 * st, 0, sizeof(double) * 10 * st, c->sc));
    c->time_kernels = (o.time_kernels & 1) != 0;
    sec_begin(c, (o.time_kernels & 2) != 0); c->sec_dump = (o.time_kernels & 4) != 0;
    c->tev_used = 0; c->spmv_calls_timed = 0;
    if (c->time_kernels && c->tev.empty()) {
        c->tev.resize(kMaxTimed);
        for (auto &e : c->tev) BICG_HIP(hipEventCreate(&e));
    }
    BICG_HIP(hipStreamSynchronize(c->sc));

    c->it = 0; c->printed = 0; c->t_iter = 0.0; c->adaptive_rr = 0;
    c->t_begin = now_sec();
    Driver d(c, method, o.krr, o.nrr);
    d.init();
    fetch_scal(c);
    c->t_init = now_sec() - c->t_begin;
}

// hipGraph replay of one iteration. The iteration body is a fixed sequence of launches (and, across
// ranks, RCCL calls on the communication stream joined back by events), identical from one
// iteration to the next except for pipe_bicgstab_rr's replacement steps, so it is captured once
// per method and replayed: one graph launch instead of 5-17 enqueue calls per iteration. This is
// what keeps an 8-GPU run (20-25 us of GPU work per iteration) from being bound by the host's
// ~3-4 us per launch. Two eager iterations come first so that every lazy initialisation (RCCL
// connections, kernel loading) happens outside the capture. Returns false when the caller has to
// run the iteration eagerly.
bool graph_iteration(bicg_ctx *c, Driver &d)
{
    const int m = c->method;
    // Off unless BICG_GRAPH=1: measured on one MI355X, eager in-order enqueueing already keeps the
    // GPU busy (54 us vs 60 us replayed per 17-op iteration of a 200 k-row rank); replay only pays
    // when the two-stream overlap mode is on (80 vs 105 us).
    const bool want = c->graph_mode == 1;
    // consumer-side finish alternates between two scalar blocks: launch arguments change from one
    // iteration to the next unless the host enqueues the collectives itself
    if (c->wave_mode && !hosted(c)) return false;
    if (!want || c->p2p || m == BICG_PIPE_BICGSTAB_RR || c->opt.rr_drift > 0.0 || c->time_kernels || c->time_sections || c->sec_exhausted ||
        !c->comm->stream_ordered()) return false;
    if (c->graph_exec[m] && c->graph_nt[m] != c->sell_nt) {     // captured with the other streaming policy
        (void)hipGraphExecDestroy(c->graph_exec[m]);
        c->graph_exec[m] = nullptr;
    }
    if (!c->graph_exec[m]) {
        if (c->graph_warm[m] < 2) { c->graph_warm[m]++; return false; }
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(c->sc, hipStreamCaptureModeThreadLocal) != hipSuccess) { c->graph_mode = 0; return false; }
        d.iterate(c->it, false, true);
        const hipError_t e = hipStreamEndCapture(c->sc, &g);
        if (e != hipSuccess || !g) {
            fprintf(stderr, "bicgstab_hip: graph capture failed (%s); continuing with eager launches\n", hipGetErrorString(e));
            (void)hipGetLastError();
            c->graph_mode = 0;
            return false;   // nothing was executed during the failed capture
        }
        const hipError_t e2 = hipGraphInstantiate(&c->graph_exec[m], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e2 != hipSuccess) { c->graph_exec[m] = nullptr; c->graph_mode = 0; return false; }
        c->graph_nt[m] = c->sell_nt;
    }
    BICG_HIP(hipGraphLaunch(c->graph_exec[m], c->sc));
    return true;
}

int run_iterate(bicg_ctx *c, int nsteps)
{
    const bicg_options &o = c->opt;
    use_device(c);
    Driver d(c, c->method, o.krr, o.nrr);
    const bool talk = c->rank == 0 && !o.quiet;
    const double t0 = now_sec();
    const int stop = std::min(o.max_iter, c->it + std::max(nsteps, 0));
    while (!c->hS->done && c->it < stop) {
        // (section marks are host-side events between launches: the multi-launch forms are what they can time)
        bool persist = c->persist_on && !c->time_kernels && !c->time_sections && !c->sec_exhausted &&
                             ((c->method >= BICG_PIPE_BICGSTAB && (c->persist.rpt == 1u || (c->method == BICG_PIPE_BICGSTAB && o.rr_drift <= 0.0))) ||
                              ((c->method == BICG_BICGSTAB || c->method == BICG_CA_BICGSTAB) && c->persist_plain && c->persist.rpt == 1u) ||
                              ((c->method == BICG_BICGSTAB || c->method == BICG_CA_BICGSTAB) && c->persist_plain && c->persist.rpt == 2u && !c->persist.mat_entries));
        // A persistent launch costs ~27 us of set-up (matrix slices and x window into LDS) and stops by itself at
        // convergence: it covers at least kPersistChunk iterations whatever the host check interval (200 k-row rank,
        // pipelined: 12.5 us per iteration at 16 per launch, 11.0 at 128, 10.9 at 512 -- tools/persist_chunk_times.py)
        const int persist_chunk_min = knob_tok("BICG_PERSIST", "chunk") ? std::max(1, atoi(knob_tok("BICG_PERSIST", "chunk"))) : kPersistChunk;
        const int chunk = std::min(persist ? std::max(o.check_every, persist_chunk_min) : o.check_every, stop - c->it);
        bool force = false;
        // (the persistent kernel checks the drift itself, every check_every iterations inside the launch)
        if (!persist && o.rr_drift > 0.0 && c->method >= BICG_PIPE_BICGSTAB && c->it > 0 && d.drift() > o.rr_drift) {
            force = true;
            c->adaptive_rr++;
        }
        sec_mark(c, SEC_VEC);
        const double t_chunk = now_sec();
        if (persist) persist = persist_chunk(c, chunk);   // one launch for the whole chunk (bicg_persist.hip); false: it could not be launched
        for (int j = 0; j < chunk && !persist; ++j) {
            // the last iteration before the caller (or the drift check) reads x / r leaves them as the reference would
            const bool last = j == chunk - 1 && (c->it + chunk >= stop || o.rr_drift > 0.0);
            if (j == 0 && force) { d.iterate(c->it, true, last); continue; }
            if (!graph_iteration(c, d)) d.iterate(c->it + j, false, last);
        }
        c->it += chunk;
        sec_mark(c, SEC_STOP);
        c->t_enq += now_sec() - t_chunk;          // (bicg_run_iterate_timed: host time until the chunk's launches were enqueued)
        fetch_scal(c);
        if (persist && c->method >= BICG_PIPE_BICGSTAB) persist_account(c);
        if (talk && o.out_iter > 0) {   // reference src/solver.c:122-126
            const int k = c->hS->k;
            const int upto = (k / o.out_iter) * o.out_iter;
            if (upto > c->printed) {
                std::vector<double> hist(k);
                BICG_HIP(hipMemcpy(hist.data(), c->trace + 3 * (size_t)c->trace_cap, sizeof(double) * k, hipMemcpyDeviceToHost));
                for (int q = c->printed + o.out_iter; q <= upto; q += o.out_iter)
                    printf("Iteration: %d, Residual: %e\n", q, sqrt(hist[q - 1] / c->hS->dot_zero));
                c->printed = upto;
            }
        }
    }
    c->t_iter += now_sec() - t0;
    return c->hS->k;
}

int run_end(bicg_ctx *c, bicg_result *res)
{
    const bicg_options &o = c->opt;
    // (a drop-in solve that lost its peer-to-peer path is about to be repeated: no summary of the aborted attempt)
    const bool talk = c->rank == 0 && !o.quiet && !(c->comm_failed && c->soft_fail);
    const int k = c->hS->k;
    c->last_iters = k;
    double spmv_ms = 0.0;
    int spmv_n = 0;
    if (c->time_kernels) {
        for (int i = 0; i + 1 < c->tev_used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->tev[i], c->tev[i + 1]) == hipSuccess) spmv_ms += ms;
        }
        spmv_n = c->spmv_calls_timed;
    }
    sec_collect(c, k);
    const double total = c->t_init + c->t_iter;
    if (res) {
        res->iterations = k;
        res->dot_r = c->hS->dot_r;
        res->dot_zero = c->hS->dot_zero;
        res->seconds = total;
        res->iter_seconds = c->t_iter;
        res->spmv_ms_total = spmv_ms;
        res->spmv_launches = spmv_n;
        res->breakdown_iteration = c->hS->breakdown_k;
        res->adaptive_replacements = c->adaptive_rr;
    }
    if (c->hS->breakdown_k && c->rank == 0 && !o.quiet)
        fprintf(stderr, "bicgstab_hip: recurrence broke down (non-finite scalar) at iteration %d\n", c->hS->breakdown_k);
    if (talk) {   // reference src/solver.c:134-141, verbatim
        printf("Total iter   : %d\n", k);
        printf("Final r      : %e\n", sqrt(c->hS->dot_r / c->hS->dot_zero));
        printf("Total time   : %e [sec.] \n", total);
        printf("Avg time/iter: %e [sec.] \n", total / k);
        fflush(stdout);
    }
    return k;
}

// Which pipelined form? By default a constant decides (fuse_small / x windows, set in bicg_create): the same program then
// takes the same form on every run, which keeps results bit-reproducible from run to run -- the two forms associate the dot
// sums differently. BICG_PLAN="pipe-probe" measures instead: the first pipelined solve on a context runs 2 + 6 iterations of each
// form on the system A x = A 1, x0 = 0 (the caller's x0 / b are restored afterwards), all ranks agree on the slower rank's times, the
// faster form stays. The extra solves advance the exchange sequence numbers: a probed solve is not bit-identical to an unprobed one
// whenever the chosen form differs from the rule's (probing trades away that reproducibility; it is opt-in).
void probe_pipe_form(bicg_ctx *c, int method, const bicg_options *opt_in)
{
    bicg_options o;
    if (opt_in) o = *opt_in; else bicg_default_options(&o);
    const bool persist = c->persist_on && method == BICG_PIPE_BICGSTAB && o.rr_drift <= 0.0 && !(o.time_kernels & 3);
    if (!c->fuse_plan_ok || hosted(c)) { c->pipe_probed = true; return; }      // this context has one form only
    if (persist || o.max_iter < 16) return;      // this solve takes the persistent form / is too short to pay for it: a later one may probe
    c->pipe_probed = true;
    use_device(c);
    const size_t n = c->n_loc;
    double *keep = dev_alloc<double>(2 * n);
    BICG_HIP(hipMemcpy(keep, c->v.x, sizeof(double) * n, hipMemcpyDeviceToDevice));
    BICG_HIP(hipMemcpy(keep + n, c->v.r, sizeof(double) * n, hipMemcpyDeviceToDevice));
    o.quiet = 1; o.tol = 0.0; o.max_iter = 8; o.check_every = 8; o.out_iter = 0; o.time_kernels = 0; o.record_trace = 0;
    // The probe does not iterate on the caller's data (an x0 that already solves the system would make the recurrences divide by
    // ~0 inside the probe): it solves A x = A 1 from x0 = 0, the reference's own test system (src/main.c:109-117). A breakdown all
    // the same counts as a tie: the rule's form stays.
    const bool rule_form = c->fuse_pipe;
    double *bsyn = dev_alloc<double>(n ? n : 1);
    {
        std::vector<double> ones(n ? n : 1, 1.0);
        scal_reset(c);
        BICG_HIP(hipMemcpyAsync(c->v.p, ones.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->sc));
        c->time_kernels = false;
        spmv(c, c->v.p, c->v.s, 0, nullptr, c->red(0, PH_NONE));
        BICG_HIP(hipMemcpyAsync(bsyn, c->v.s, sizeof(double) * n, hipMemcpyDeviceToDevice, c->sc));
        BICG_HIP(hipStreamSynchronize(c->sc));
    }
    double t[2] = {0.0, 0.0};
    bool broke = false;
    for (int form = 1; form >= 0; --form) {
        c->fuse_pipe = form != 0;
        BICG_HIP(hipMemset(c->v.x, 0, sizeof(double) * n));
        BICG_HIP(hipMemcpy(c->v.r, bsyn, sizeof(double) * n, hipMemcpyDeviceToDevice));
        run_begin(c, method, &o);
        run_iterate(c, 2);
        const double t0 = now_sec();
        run_iterate(c, 6);
        t[form] = (now_sec() - t0) / 6.0 * 1.0e3;
        broke = broke || c->hS->breakdown_k != 0 || c->hS->comm_error != 0;
    }
    BICG_HIP(hipFree(bsyn));
    if (c->nranks > 1) {      // the slower rank's time counts, and every rank must take the same decision
        const int P = c->nranks;
        std::vector<int> cnt(P, 2 * (int)sizeof(double)), off(P);
        std::vector<double> mine(2 * (size_t)P), all(2 * (size_t)P, 0.0);
        for (int p = 0; p < P; ++p) { off[p] = 2 * p * (int)sizeof(double); mine[2 * p] = t[0]; mine[2 * p + 1] = t[1]; }
        c->comm->alltoallv_host(mine.data(), cnt.data(), off.data(), all.data(), cnt.data(), off.data());
        all[2 * c->rank] = t[0]; all[2 * c->rank + 1] = t[1];
        for (int p = 0; p < P; ++p) { t[0] = std::max(t[0], all[2 * p]); t[1] = std::max(t[1], all[2 * p + 1]); }
    }
    c->probe_ms[0] = t[0]; c->probe_ms[1] = t[1];
    if (c->nranks > 1) broke = !all_ranks(c->comm, !broke);
    c->fuse_pipe = broke ? rule_form : t[1] <= t[0];
    BICG_HIP(hipMemcpy(c->v.x, keep, sizeof(double) * n, hipMemcpyDeviceToDevice));
    BICG_HIP(hipMemcpy(c->v.r, keep + n, sizeof(double) * n, hipMemcpyDeviceToDevice));
    BICG_HIP(hipFree(keep));
    if (c->rank == 0 && knob_x("BICG_PIPE_PROBE_VERBOSE"))
        fprintf(stderr, "bicgstab_hip: pipelined form probe: separate kernels %.4f ms, SpMV epilogues %.4f ms per iteration -> %s\n", t[0], t[1],
                c->fuse_pipe ? "epilogues" : "separate kernels");
}

int run_solver(bicg_ctx *c, int method, const bicg_options *opt_in, bicg_result *res)
{
    if (c->pipe_probe && !c->pipe_probed && method >= BICG_PIPE_BICGSTAB) probe_pipe_form(c, method, opt_in);
    run_begin(c, method, opt_in);
    run_iterate(c, c->opt.max_iter);
    return run_end(c, res);
}

