// bicg_comm.h -- rank-to-rank transports behind the solver (one process per GPU).
//
// The reference communicates through MPI_COMM_WORLD only: MPI_Iallgatherv of the whole vector per
// SpMV (src/matrix.c:432) and one 8-byte MPI_Iallreduce per dot product (e.g. src/solver.c:90).
// Here: a halo exchange of exactly the entries the offd block references, and ONE packed
// all-reduce per dot group, on RCCL over xGMI (device buffers, enqueued on a HIP stream) or on a
// host-staged transport (callbacks: MPI, gloo, ...) used for tests and for ranks > GPUs.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/bicgstab_hip.h"
#include "bicg_device.h"

namespace bicg {

struct P2p;

struct Comm {
    int rank = 0, nranks = 1, device = 0;
    // Direct peer-to-peer data path (bicg_p2p.cpp), layered on top of any transport once
    // p2p_enable() has succeeded on every rank: halo values and dot sums are then stored straight
    // into the other GPUs' memory by the producing kernels; the transport below only bootstraps.
    P2p *p2p = nullptr;
    // the peer-to-peer path was switched on by "auto" selection (bicg_comm_init_mpi), not asked for: a time-out in a
    // drop-in solve then falls back to this transport's own collectives instead of ending the program
    bool p2p_auto = false;
    // ranks of this communicator that drive the SAME GPU as this one (1 in production; > 1 when a one-GPU box
    // stands in for a node in tests). Kernels that wait for another rank's data hold their workgroup slots while
    // they wait; with several ranks on one device the launches of all of them must fit on it together, or the
    // waiting workgroups of one rank keep out the workgroups of the rank they are waiting for (p2p_enable sets it).
    int ranks_on_device = 1;
    virtual ~Comm();
    virtual const char *name() const = 0;
    // true: collectives are enqueued on `st` and complete in stream order (RCCL);
    // false: the call synchronises `st` and completes on return (host staged)
    virtual bool stream_ordered() const = 0;
    // in-place sum over ranks of n doubles in DEVICE memory
    virtual void allreduce_sum(double *dev, int n, hipStream_t st) = 0;
    // halo exchange, DEVICE buffers; counts/displs in doubles per peer rank
    virtual void exchange(const double *send, const int *scnt, const int *sdsp, double *recv, const int *rcnt,
                          const int *rdsp, hipStream_t st) = 0;
    // set-up time personalised exchange of HOST bytes
    virtual void alltoallv_host(const void *send, const int *scnt, const int *sdsp, void *recv, const int *rcnt,
                                const int *rdsp) = 0;
};

// Peer-to-peer transport over xGMI (or within one GPU for tests): every rank maps the other
// ranks' mailboxes through HIP IPC.
struct P2p {
    Comm *comm = nullptr;
    int rank = 0, nranks = 1;
    bool uncached = false;                 // mailboxes live in uncached (fine-grained) device memory
    llword *mail = nullptr;                // this rank's all-reduce mailbox
    llword **mail_dev = nullptr;           // device array [nranks]: all mailboxes as mapped here
    std::vector<void *> mapped;            // IPC mappings of the mailboxes
    unsigned red_seq = 1;                  // sequence number of the next all-reduce group (0 is never used)
    unsigned bar_seq = 1;                  // ... of the next barrier token
    unsigned long long timeout_ticks = 0;  // 100 MHz ticks a kernel waits for a peer before giving up

    P2pRed red_desc(unsigned seq, unsigned mask = 0xffu) const
    {
        P2pRed r;
        r.mail = mail_dev; r.seq = seq; r.mask = mask; r.rank = rank; r.nranks = nranks;
        r.n_collect = 0; r.timeout_ticks = timeout_ticks;
        return r;
    }
    void *alloc(size_t bytes);             // zero-filled UNCACHED device memory other ranks may map; nullptr if unavailable
    void release(void *p);
    // Collective. Maps `local` of every rank into this process: peers[p] (peers[rank] = local);
    // the mappings of the other ranks are appended to `opened`. 0 when it worked on EVERY rank.
    int share(void *local, std::vector<void *> &peers, std::vector<void *> &opened);
    void unmap(std::vector<void *> &opened);
    ~P2p();
};
// Collective: set up and self-test the peer-to-peer path on communicator c. 0 = enabled on every
// rank; otherwise nothing changed and the transport keeps working as before.
int p2p_enable(Comm *c);
// Collective: give the peer-to-peer path up again (no context may still be using it).
void p2p_disable(Comm *c);

Comm *comm_get();                 // process-global communicator (auto-initialised on first use)
void comm_set(Comm *c);           // takes ownership
void contexts_orphan();           // bicg_create.cpp: contexts built on the communicator that is about to go away
Comm *make_single(int device);
Comm *make_host(int rank, int nranks, bicg_allreduce_fn ar, bicg_alltoallv_fn a2a, void *user, int device);
Comm *make_rccl(int rank, int nranks, const void *id, int device);
int   rccl_unique_id(void *out);
int   rccl_loadable();
int   pick_device(int rank, int requested);

[[noreturn]] void die(const char *what, const char *detail);
#define BICG_HIP(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) ::bicg::die(#call, hipGetErrorString(e_));                     \
    } while (0)

}  // namespace bicg

// MPI through weak symbols (bicg_mpi_shim.c)
extern "C" {
int  bicg_mpi_active(void);
void bicg_mpi_rank_size(int *rank, int *size);
void bicg_mpi_bcast_bytes(void *buf, int n, int root);
void bicg_mpi_allreduce_sum(double *buf, int n, void *user);
void bicg_mpi_alltoallv_bytes(const void *send, const int *scnt, const int *sdsp, void *recv, const int *rcnt,
                              const int *rdsp, void *user);
}
