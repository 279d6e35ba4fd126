// bicg_comm.cpp -- transports: single rank, RCCL over xGMI (dlopen'ed, one process per GPU),
// host-staged callbacks. See bicg_comm.h.
#include "bicg_comm.h"
#include "bicg_parallel.h"
#include "bicg_knobs.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only; the functions are resolved with dlsym

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace bicg {

void die(const char *what, const char *detail)
{
    // the reference's loader reports errors this way (src/matrix.c:34-37)
    fprintf(stderr, "ERROR: bicgstab_hip: %s: %s\n", what, detail ? detail : "");
    fflush(stderr);
    exit(EXIT_FAILURE);
}

Comm::~Comm() { delete p2p; }

int pick_device(int rank, int requested)
{
    int ndev = 0;
    BICG_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) die("hipGetDeviceCount", "no HIP device visible");
    int dev = requested;
    if (dev < 0) {
        const char *lr = getenv("LOCAL_RANK");
        dev = (lr ? atoi(lr) : rank) % ndev;
    }
    BICG_HIP(hipSetDevice(dev));
    return dev;
}

// ------------------------------------------------------------------ single rank
struct SingleComm : Comm {
    const char *name() const override { return "single"; }
    bool stream_ordered() const override { return true; }
    void allreduce_sum(double *, int, hipStream_t) override {}
    void exchange(const double *, const int *, const int *, double *, const int *, const int *, hipStream_t) override {}
    void alltoallv_host(const void *, const int *, const int *, void *, const int *, const int *) override {}
};

Comm *make_single(int device)
{
    SingleComm *c = new SingleComm;
    c->rank = 0; c->nranks = 1;
    c->device = pick_device(0, device);
    return c;
}

// ------------------------------------------------------------------ host staged
struct HostComm : Comm {
    bicg_allreduce_fn ar = nullptr;
    bicg_alltoallv_fn a2a = nullptr;
    void *user = nullptr;
    std::vector<double> hs, hr;
    std::vector<int> bs_cnt, bs_dsp, br_cnt, br_dsp;

    const char *name() const override { return "host"; }
    bool stream_ordered() const override { return false; }

    void allreduce_sum(double *dev, int n, hipStream_t st) override
    {
        double tmp[64];
        if (n > 64) die("allreduce_sum", "group too wide");
        BICG_HIP(hipMemcpyAsync(tmp, dev, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        BICG_HIP(hipStreamSynchronize(st));
        ar(tmp, n, user);
        BICG_HIP(hipMemcpyAsync(dev, tmp, sizeof(double) * n, hipMemcpyHostToDevice, st));
        BICG_HIP(hipStreamSynchronize(st));
    }

    void exchange(const double *send, const int *scnt, const int *sdsp, double *recv, const int *rcnt, const int *rdsp,
                  hipStream_t st) override
    {
        const int P = nranks;
        const int ns = sdsp[P - 1] + scnt[P - 1], nr = rdsp[P - 1] + rcnt[P - 1];
        hs.resize(ns > 0 ? ns : 1); hr.resize(nr > 0 ? nr : 1);
        bs_cnt.resize(P); bs_dsp.resize(P); br_cnt.resize(P); br_dsp.resize(P);
        for (int p = 0; p < P; ++p) {
            bs_cnt[p] = scnt[p] * 8; bs_dsp[p] = sdsp[p] * 8; br_cnt[p] = rcnt[p] * 8; br_dsp[p] = rdsp[p] * 8;
        }
        if (ns) BICG_HIP(hipMemcpyAsync(hs.data(), send, sizeof(double) * ns, hipMemcpyDeviceToHost, st));
        BICG_HIP(hipStreamSynchronize(st));
        a2a(hs.data(), bs_cnt.data(), bs_dsp.data(), hr.data(), br_cnt.data(), br_dsp.data(), user);
        if (nr) BICG_HIP(hipMemcpyAsync(recv, hr.data(), sizeof(double) * nr, hipMemcpyHostToDevice, st));
        BICG_HIP(hipStreamSynchronize(st));
    }

    void alltoallv_host(const void *send, const int *scnt, const int *sdsp, void *recv, const int *rcnt,
                        const int *rdsp) override
    {
        a2a(send, scnt, sdsp, recv, rcnt, rdsp, user);
    }
};

Comm *make_host(int rank, int nranks, bicg_allreduce_fn ar, bicg_alltoallv_fn a2a, void *user, int device)
{
    HostComm *c = new HostComm;
    c->rank = rank; c->nranks = nranks; c->ar = ar; c->a2a = a2a; c->user = user;
    c->device = pick_device(rank, device);
    return c;
}

// ------------------------------------------------------------------ RCCL (xGMI)
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    const char *(*GetLastError)(ncclComm_t) = nullptr;       // optional (RCCL >= 2.13): the library's own text of what went wrong
};
static char g_rccl_error[512] = "";     // why the last ncclCommInitRank of this process failed (bicg_comm_last_error)

static RcclApi &rccl()
{
    static RcclApi api;
    if (api.handle) return api;
    // librccl.so.1 resolves to an already loaded RCCL (e.g. the one PyTorch ships) by SONAME.
    // RTLD_LOCAL: RCCL pulls in /opt/rocm's librocm_smi64; made global, a PyTorch imported LATER binds the statics of
    // its own bundled librocm_smi64 to that copy and both destroy them at exit ("double free or corruption", exit 134
    // after every test had passed)
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
    }
    if (!api.handle) die("dlopen(librccl)", dlerror());
#define SYM(field, sym)                                                       \
    do {                                                                      \
        *(void **)(&api.field) = dlsym(api.handle, sym);                      \
        if (!api.field) die("dlsym", sym);                                    \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *(void **)(&api.GetLastError) = dlsym(api.handle, "ncclGetLastError");
    return api;
}

#define BICG_NCCL(call)                                                      \
    do {                                                                     \
        ncclResult_t r_ = (call);                                            \
        if (r_ != ncclSuccess) die(#call, rccl().GetErrorString(r_));        \
    } while (0)

int rccl_loadable() { return rccl().handle != nullptr; }

int rccl_unique_id(void *out)
{
    static_assert(sizeof(ncclUniqueId) == BICG_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    BICG_NCCL(rccl().GetUniqueId(&id));
    memcpy(out, &id, sizeof id);
    return 0;
}

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;

    ~RcclComm() override { if (comm) rccl().CommDestroy(comm); }
    const char *name() const override { return "rccl"; }
    bool stream_ordered() const override { return true; }

    void allreduce_sum(double *dev, int n, hipStream_t st) override
    {
        BICG_NCCL(rccl().AllReduce(dev, dev, (size_t)n, ncclFloat64, ncclSum, comm, st));
    }

    // point-to-point to the actual neighbours only: on xGMI every pair of GPUs has a direct link
    void exchange(const double *send, const int *scnt, const int *sdsp, double *recv, const int *rcnt, const int *rdsp,
                  hipStream_t st) override
    {
        BICG_NCCL(rccl().GroupStart());
        for (int p = 0; p < nranks; ++p) {
            if (p == rank) continue;
            if (scnt[p] > 0) BICG_NCCL(rccl().Send(send + sdsp[p], (size_t)scnt[p], ncclFloat64, p, comm, st));
            if (rcnt[p] > 0) BICG_NCCL(rccl().Recv(recv + rdsp[p], (size_t)rcnt[p], ncclFloat64, p, comm, st));
        }
        BICG_NCCL(rccl().GroupEnd());
    }

    void alltoallv_host(const void *send, const int *scnt, const int *sdsp, void *recv, const int *rcnt,
                        const int *rdsp) override
    {
        const int P = nranks;
        const size_t ns = (size_t)sdsp[P - 1] + scnt[P - 1], nr = (size_t)rdsp[P - 1] + rcnt[P - 1];
        char *ds = nullptr, *dr = nullptr;
        BICG_HIP(hipMalloc((void **)&ds, ns ? ns : 1));
        BICG_HIP(hipMalloc((void **)&dr, nr ? nr : 1));
        if (ns) BICG_HIP(hipMemcpy(ds, send, ns, hipMemcpyHostToDevice));
        // the self segment never travels
        if (scnt[rank] > 0) memcpy((char *)recv + rdsp[rank], (const char *)send + sdsp[rank], (size_t)scnt[rank]);
        hipStream_t st = nullptr;
        BICG_NCCL(rccl().GroupStart());
        for (int p = 0; p < P; ++p) {
            if (p == rank) continue;
            if (scnt[p] > 0) BICG_NCCL(rccl().Send(ds + sdsp[p], (size_t)scnt[p], ncclInt8, p, comm, st));
            if (rcnt[p] > 0) BICG_NCCL(rccl().Recv(dr + rdsp[p], (size_t)rcnt[p], ncclInt8, p, comm, st));
        }
        BICG_NCCL(rccl().GroupEnd());
        BICG_HIP(hipStreamSynchronize(st));
        for (int p = 0; p < P; ++p)
            if (p != rank && rcnt[p] > 0)
                BICG_HIP(hipMemcpy((char *)recv + rdsp[p], dr + rdsp[p], (size_t)rcnt[p], hipMemcpyDeviceToHost));
        BICG_HIP(hipFree(ds));
        BICG_HIP(hipFree(dr));
    }
};

Comm *make_rccl(int rank, int nranks, const void *id, int device)
{
    RcclComm *c = new RcclComm;
    c->rank = rank; c->nranks = nranks;
    c->device = pick_device(rank, device);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, uid, rank);
    g_rccl_error[0] = 0;
    if (r != ncclSuccess) {
        const char *last = rccl().GetLastError ? rccl().GetLastError(nullptr) : nullptr;
        snprintf(g_rccl_error, sizeof g_rccl_error, "ncclCommInitRank(nranks %d, rank %d, device %d): %s%s%s", nranks, rank, c->device,
                 rccl().GetErrorString(r), last && last[0] ? " -- " : "", last && last[0] ? last : "");
        // BICG_COMM_SOFT_FAIL=1 (bench.py): report instead of exiting, so that the caller can fall back to
        // another transport collectively
        const char *soft = getenv("BICG_COMM_SOFT_FAIL");
        if (!soft || atoi(soft) == 0) die("ncclCommInitRank", rccl().GetErrorString(r));
        fprintf(stderr, "bicgstab_hip: rank %d: ncclCommInitRank failed: %s\n", rank, rccl().GetErrorString(r));
        c->comm = nullptr;
        delete c;
        return nullptr;
    }
    return c;
}

// ------------------------------------------------------------------ process-global communicator
static Comm *g_comm = nullptr;

void comm_set(Comm *c)
{
    if (g_comm) {
        bicg_dropin_release();       // the resident drop-in context was built on this communicator
        contexts_orphan();           // contexts the caller still holds: they give back what lives in the transport
        delete g_comm;
    }
    g_comm = c;
    plan_ranks_hint() = c ? c->nranks : 1;      // set-up threads per rank: the host's threads are shared (bicg_parallel.h)
}

// First use without an explicit bicg_comm_init_*: adopt MPI_COMM_WORLD when the host program has
// initialised MPI (the reference's main.c does, src/main.c:14), else run single rank.
Comm *comm_get()
{
    if (g_comm) return g_comm;
    if (bicg_mpi_active()) {
        const char *t = getenv("BICG_TRANSPORT");
        if (bicg_comm_init_mpi(t ? t : "auto", -1) == 0) return g_comm;
    }
    g_comm = make_single(-1);
    return g_comm;
}

}  // namespace bicg

using namespace bicg;

extern "C" {

int bicg_comm_unique_id(void *id_out) { return rccl_unique_id(id_out); }
int bicg_comm_rccl_loadable(void) { return rccl_loadable(); }
int bicg_comm_last_error(char *out, int cap)
{
    if (out && cap > 0) { strncpy(out, g_rccl_error, (size_t)cap - 1); out[cap - 1] = 0; }
    return (int)strlen(g_rccl_error);
}

int bicg_comm_init_rccl(int rank, int nranks, const void *id, int device)
{
    const char *force = test_tok("force-comm");     // tests: a real 1-rank RCCL communicator
    Comm *c = nranks > 1 || (force && atoi(force)) ? make_rccl(rank, nranks, id, device) : make_single(device);
    if (!c) return 1;                                  // BICG_COMM_SOFT_FAIL: no communicator was installed
    comm_set(c);
    return 0;
}

int bicg_comm_init_host(int rank, int nranks, bicg_allreduce_fn allreduce, bicg_alltoallv_fn alltoallv, void *user,
                        int device)
{
    comm_set(nranks > 1 ? make_host(rank, nranks, allreduce, alltoallv, user, device) : make_single(device));
    return 0;
}

int bicg_comm_init_single(int device)
{
    comm_set(make_single(device));
    return 0;
}

int bicg_comm_init_mpi(const char *transport, int device)
{
    if (!bicg_mpi_active()) return 1;
    int rank = 0, size = 1;
    bicg_mpi_rank_size(&rank, &size);
    if (size == 1) { comm_set(make_single(device)); return 0; }
    int ndev = 0;
    BICG_HIP(hipGetDeviceCount(&ndev));
    // "auto": RCCL needs one GPU per rank, otherwise the exchanges are staged through MPI on the host;
    //         with one GPU per rank the data path then moves to direct peer-to-peer stores when the
    //         self-test passes (bicg_p2p.cpp). "rccl" / "host": exactly that transport.
    //         "p2p": peer-to-peer on top of whatever transport fits, also when ranks share a GPU.
    const bool want_p2p = transport && strcmp(transport, "p2p") == 0;
    bool use_rccl;
    if (transport && strcmp(transport, "rccl") == 0) use_rccl = true;
    else if (transport && strcmp(transport, "host") == 0) use_rccl = false;
    else use_rccl = size <= ndev;
    if (use_rccl) {
        char id[BICG_UNIQUE_ID_BYTES];
        if (rank == 0) rccl_unique_id(id);
        bicg_mpi_bcast_bytes(id, BICG_UNIQUE_ID_BYTES, 0);
        Comm *rc = make_rccl(rank, size, id, device);
        // BICG_COMM_SOFT_FAIL: a rank whose ncclCommInitRank failed gets nullptr. The transport must be the same
        // everywhere (ranks on different transports hang in their first collective), so the verdicts are summed over
        // MPI and ONE failure sends every rank to the MPI-staged transport.
        double failed = rc ? 0.0 : 1.0;
        bicg_mpi_allreduce_sum(&failed, 1, nullptr);
        if (failed == 0.0) {
            comm_set(rc);
        } else {
            if (rc && rank == 0)
                fprintf(stderr, "bicgstab_hip: %d rank(s) could not join the RCCL communicator; all ranks use the MPI-staged transport\n", (int)failed);
            delete rc;
            use_rccl = false;
        }
    }
    if (!use_rccl) {
        comm_set(make_host(rank, size, bicg_mpi_allreduce_sum, bicg_mpi_alltoallv_bytes, nullptr, device));
    }
    const bool automatic = !transport || strcmp(transport, "auto") == 0;
    if (want_p2p || (automatic && use_rccl)) {
        const int rc = p2p_enable(g_comm);
        // a bet on the self-test: the drop-in path can take it back (BICG_P2P_FALLBACK=1: also when asked for explicitly)
        const char *fb = getenv("BICG_P2P_FALLBACK");
        if (rc == 0 && (!want_p2p || (fb && atoi(fb)))) g_comm->p2p_auto = true;
        if (rc != 0 && want_p2p && rank == 0)
            fprintf(stderr, "bicgstab_hip: peer-to-peer transport not available (code %d), using %s\n", rc, g_comm->name());
    }
    return 0;
}

// One-rank RCCL round trip on the current device: dlopen + symbol resolution, the by-value
// ncclUniqueId ABI, communicator creation, an in-place fp64 all-reduce and a grouped (empty)
// exchange on a non-default stream. Returns 0 when the reduced values come back unchanged.
int bicg_comm_selftest_rccl(int device)
{
    char id[BICG_UNIQUE_ID_BYTES];
    rccl_unique_id(id);
    Comm *c = make_rccl(0, 1, id, device);
    if (!c) return 100;                // BICG_COMM_SOFT_FAIL and the communicator could not be created
    hipStream_t st;
    BICG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double h[5] = {1.5, -2.0, 3.25, 0.0, 1e300}, back[5] = {0, 0, 0, 0, 0};
    double *d = nullptr;
    BICG_HIP(hipMalloc((void **)&d, sizeof h));
    BICG_HIP(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    c->allreduce_sum(d, 5, st);
    int zero = 0;
    c->exchange(d, &zero, &zero, d, &zero, &zero, st);
    BICG_HIP(hipStreamSynchronize(st));
    BICG_HIP(hipMemcpy(back, d, sizeof h, hipMemcpyDeviceToHost));
    BICG_HIP(hipFree(d));
    BICG_HIP(hipStreamDestroy(st));
    delete c;
    for (int i = 0; i < 5; ++i) if (back[i] != h[i]) return 1 + i;
    return 0;
}

void bicg_comm_finalize(void) { comm_set(nullptr); }
int bicg_comm_rank(void) { return comm_get()->rank; }
int bicg_comm_size(void) { return comm_get()->nranks; }

}  // extern "C"
