// bicg_p2p.cpp -- direct peer-to-peer data path between the GPUs of one node.
//
// The reference exchanges a whole vector (MPI_Iallgatherv, src/matrix.c:432) and one double per dot
// product (MPI_Iallreduce, e.g. src/solver.c:90) through MPI. On one MI355X node every pair of GPUs
// shares a direct xGMI link, and a BiCGStab iteration of a 200 k-row rank is ~20 us of GPU work, so
// a library collective per exchange (15-25 us each, five per iteration) would dominate it. Here the
// kernels that PRODUCE a value store it straight into the memory of the GPUs that need it:
//
//   * every rank allocates a small mailbox (all-reduce) and, per matrix, a halo landing ring in
//     uncached device memory and publishes them through HIP IPC; this file does that set-up, once;
//   * values travel as LL words (bicg_device.h): payload and sequence number in one 8-byte store,
//     so no fence, flag or acknowledgement is needed and a reader can never see a torn value;
//   * waits happen inside small kernels with a wall-clock bound: a missing peer produces an error
//     (Scal::comm_error), never a hung GPU.
//
// The transport underneath (RCCL, MPI or caller-supplied callbacks) is only used for the set-up
// exchanges of IPC handles. p2p_enable() finishes with a self-test on the real links and every
// rank learns whether ALL ranks passed; if not, the path is left disabled and the solver keeps
// using the transport's own collectives.
#include "bicg_comm.h"
#include "bicg_knobs.h"

#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bicg {

namespace {

struct ShareMsg {
    int ok;                     // the sender could export its buffer
    int device;
    unsigned long long host;    // hash of the host name: IPC handles only mean something on one node
    long long pid;
    hipIpcMemHandle_t handle;
};

unsigned long long host_hash()
{
    char name[256];
    memset(name, 0, sizeof name);
    (void)gethostname(name, sizeof name - 1);
    unsigned long long h = 1469598103934665603ull;
    for (const char *p = name; *p; ++p) h = (h ^ (unsigned char)*p) * 1099511628211ull;
    return h;
}

// every rank contributes one int; true when all of them are zero
bool all_zero(Comm *c, int mine)
{
    const int P = c->nranks;
    std::vector<int> cnt(P, (int)sizeof(int)), dsp(P), out(P, mine), in(P, 0);
    for (int p = 0; p < P; ++p) dsp[p] = p * (int)sizeof(int);
    c->alltoallv_host(out.data(), cnt.data(), dsp.data(), in.data(), cnt.data(), dsp.data());
    in[c->rank] = mine;
    for (int p = 0; p < P; ++p)
        if (in[p] != 0) return false;
    return true;
}

}  // namespace

void *P2p::alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    // Mailboxes and landing rings are written by other GPUs and polled here: they must live in UNCACHED
    // (fine-grained) device memory. With ordinary hipMalloc memory a line polled earlier may stay in this
    // GPU's L2, which a peer's store over xGMI does not invalidate -- a reused ring slot could be read
    // stale until the time-out. No silent fallback: if the runtime cannot provide (or export) uncached
    // memory the peer-to-peer path stays off and the transport's own collectives are used.
    // BICG_P2P_ALLOC=default asks for ordinary memory explicitly (single-GPU experiments only).
    const char *mode = knob_x("BICG_P2P_ALLOC");
    uncached = false;
    if (mode && strcmp(mode, "default") == 0) {
        BICG_HIP(hipMalloc(&p, bytes));
    } else if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess && p) {
        uncached = true;
    } else {
        (void)hipGetLastError();
        return nullptr;
    }
    BICG_HIP(hipMemset(p, 0, bytes));
    BICG_HIP(hipDeviceSynchronize());
    return p;
}

void P2p::release(void *p)
{
    if (p) (void)hipFree(p);
}

int P2p::share(void *local, std::vector<void *> &peers, std::vector<void *> &opened)
{
    const int P = nranks;
    peers.assign(P, nullptr);
    peers[rank] = local;
    ShareMsg mine;
    memset(&mine, 0, sizeof mine);
    mine.device = comm->device;
    mine.host = host_hash();
    mine.pid = (long long)getpid();
    mine.ok = local && hipIpcGetMemHandle(&mine.handle, local) == hipSuccess ? 1 : 0;
    if (!mine.ok) (void)hipGetLastError();

    std::vector<ShareMsg> out(P, mine), in(P);
    std::vector<int> cnt(P, (int)sizeof(ShareMsg)), dsp(P);
    for (int p = 0; p < P; ++p) dsp[p] = p * (int)sizeof(ShareMsg);
    comm->alltoallv_host(out.data(), cnt.data(), dsp.data(), in.data(), cnt.data(), dsp.data());

    int bad = mine.ok ? 0 : 1;
    for (int p = 0; p < P && !bad; ++p) {
        if (p == rank) continue;
        if (!in[p].ok || in[p].host != mine.host) { bad = 1; break; }
        if (in[p].pid == mine.pid) { bad = 1; break; }      // one process per rank
        int ndev = 0, can = 1;
        if (hipGetDeviceCount(&ndev) == hipSuccess && in[p].device != mine.device && in[p].device < ndev &&
            hipDeviceCanAccessPeer(&can, mine.device, in[p].device) == hipSuccess && !can) { bad = 1; break; }   // no link to that GPU
        void *ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, in[p].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !ptr) {
            (void)hipGetLastError();
            bad = 1;
            break;
        }
        peers[p] = ptr;
        opened.push_back(ptr);
    }
    const bool ok = all_zero(comm, bad);
    return ok ? 0 : 1;
}

void P2p::unmap(std::vector<void *> &opened)
{
    for (void *p : opened) (void)hipIpcCloseMemHandle(p);
    opened.clear();
}

P2p::~P2p()
{
    unmap(mapped);
    if (mail_dev) (void)hipFree(mail_dev);
    release(mail);
}

int p2p_enable(Comm *c)
{
    if (c->p2p) return 0;
    if (const char *e = getenv("BICG_P2P")) if (atoi(e) == 0) return 1;
    if (c->nranks > kMaxRanksP2p) return 1;
    if (strcmp(c->name(), "single") == 0) return 0;     // nothing to exchange
    BICG_HIP(hipSetDevice(c->device));

    {   // who shares my GPU? (host name hash, device ordinal) of every rank
        const int P = c->nranks;
        struct Where { unsigned long long host; int device; int pad; } mine = {host_hash(), c->device, 0};
        std::vector<Where> out(P, mine), in(P);
        std::vector<int> cnt(P, (int)sizeof(Where)), dsp(P);
        for (int p = 0; p < P; ++p) dsp[p] = p * (int)sizeof(Where);
        c->alltoallv_host(out.data(), cnt.data(), dsp.data(), in.data(), cnt.data(), dsp.data());
        in[c->rank] = mine;
        int same = 0;
        for (int p = 0; p < P; ++p) same += in[p].host == mine.host && in[p].device == mine.device;
        c->ranks_on_device = same > 0 ? same : 1;
    }
    P2p *t = new P2p;
    t->comm = c; t->rank = c->rank; t->nranks = c->nranks;
    // a rank may legitimately arrive late at a collective solver call (I/O, printing): wait long (10 s; bench.py: 8 s)
    // before declaring a peer lost -- the spinning kernels occupy one wavefront each
    double timeout_ms = 10000.0;
    if (const char *e = getenv("BICG_P2P_TIMEOUT_MS")) timeout_ms = atof(e);
    t->timeout_ticks = (unsigned long long)(timeout_ms * 1.0e5);     // 100 MHz wall clock

    const size_t words = (size_t)kMailRing * c->nranks * kRedSlots * 2;
    t->mail = (llword *)t->alloc(words * sizeof(llword));
    std::vector<void *> peers;
    int rc = t->share(t->mail, peers, t->mapped);
    if (rc == 0) {
        BICG_HIP(hipMalloc((void **)&t->mail_dev, sizeof(llword *) * c->nranks));
        BICG_HIP(hipMemcpy(t->mail_dev, peers.data(), sizeof(llword *) * c->nranks, hipMemcpyHostToDevice));

        // self-test on the real links: 64 all-reduces of known values
        const int rounds = 64;
        int *status = nullptr, h[2] = {0, 0};
        BICG_HIP(hipMalloc((void **)&status, sizeof h));
        BICG_HIP(hipMemset(status, 0, sizeof h));
        hipStream_t st;
        BICG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        BICG_HIP(hipDeviceSynchronize());
        launch_p2p_selftest(t->red_desc(t->red_seq), t->red_seq, rounds, t->timeout_ticks, status, st);
        t->red_seq += rounds;
        BICG_HIP(hipStreamSynchronize(st));
        BICG_HIP(hipMemcpy(h, status, sizeof h, hipMemcpyDeviceToHost));
        BICG_HIP(hipStreamDestroy(st));
        BICG_HIP(hipFree(status));
        int bad = (h[0] != 0 || h[1] != 0) ? 1 : 0;
        if (bad && !getenv("BICG_QUIET"))
            fprintf(stderr, "bicgstab_hip: rank %d: peer-to-peer self-test failed (%d wrong sums, %d time-outs)\n", c->rank, h[0], h[1]);
        rc = all_zero(c, bad) ? 0 : 2;
        if (rc == 0) {
            // halo pattern: landing rings of the solver's kind, more rounds than slots, ranks out of step
            const int entries = 64, rounds = 3 * kHaloRing + 2;
            llword *ring = (llword *)t->alloc(sizeof(llword) * 2 * (size_t)kHaloRing * c->nranks * entries);
            std::vector<void *> rpeers, ropened;
            int rrc = t->share(ring, rpeers, ropened);
            if (rrc == 0) {
                llword **rings_dev = nullptr;
                BICG_HIP(hipMalloc((void **)&rings_dev, sizeof(llword *) * c->nranks));
                BICG_HIP(hipMemcpy(rings_dev, rpeers.data(), sizeof(llword *) * c->nranks, hipMemcpyHostToDevice));
                int *st2 = nullptr, h2[2] = {0, 0};
                BICG_HIP(hipMalloc((void **)&st2, sizeof h2));
                BICG_HIP(hipMemset(st2, 0, sizeof h2));
                hipStream_t s2;
                BICG_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
                BICG_HIP(hipDeviceSynchronize());
                launch_p2p_ringtest(t->red_desc(0), rings_dev, entries, 1u, rounds, t->bar_seq, t->timeout_ticks, st2, s2);
                t->bar_seq += (unsigned)(rounds / (kHaloRing - 2) + 1);
                BICG_HIP(hipStreamSynchronize(s2));
                BICG_HIP(hipMemcpy(h2, st2, sizeof h2, hipMemcpyDeviceToHost));
                BICG_HIP(hipStreamDestroy(s2));
                BICG_HIP(hipFree(st2));
                BICG_HIP(hipFree(rings_dev));
                bad = (h2[0] != 0 || h2[1] != 0) ? 1 : 0;
                if (bad && !getenv("BICG_QUIET"))
                    fprintf(stderr, "bicgstab_hip: rank %d: peer-to-peer halo-ring self-test failed (%d stale or wrong values, %d time-outs)\n",
                            c->rank, h2[0], h2[1]);
            } else {
                bad = 1;
            }
            rc = all_zero(c, bad) ? 0 : 3;     // every rank has finished with the test rings before any is unmapped
            t->unmap(ropened);
            t->release(ring);
        }
    }
    if (rc != 0) {
        delete t;
        return rc;
    }
    c->p2p = t;
    return 0;
}

void p2p_disable(Comm *c)
{
    if (!c->p2p) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    delete c->p2p;
    c->p2p = nullptr;
    c->p2p_auto = false;
}

}  // namespace bicg

extern "C" int bicg_comm_enable_p2p(void) { return bicg::p2p_enable(bicg::comm_get()); }

extern "C" int bicg_comm_p2p_active(void)
{
    bicg::Comm *c = bicg::comm_get();
    return c->p2p ? (c->p2p->uncached ? 2 : 1) : 0;
}
