"""Host-staged transport callbacks on top of torch.distributed (gloo): lets several ranks drive
libbicgstab_hip.so without RCCL -- used by the world_size-2 tests (CPU: plan logic; GPU box: two
ranks sharing the one GPU) exactly where the reference would use MPI (MPI_Iallreduce,
MPI_Iallgatherv; reference src/solver.c:90, src/matrix.c:432)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import hipsolver as H

_keep = []


def _bytes_view(addr, nbytes):
    if nbytes == 0 or not addr:
        return torch.empty(0, dtype=torch.uint8)
    buf = (C.c_ubyte * nbytes).from_address(addr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))


def callbacks():
    rank, world = dist.get_rank(), dist.get_world_size()

    def allreduce(buf, n, user):
        arr = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(arr)
        dist.all_reduce(t)            # in place: t shares memory with the caller's buffer

    def alltoallv(send, scnt, sdsp, recv, rcnt, rdsp, user):
        reqs, stash = [], []
        for p in range(world):
            if p == rank:
                continue
            if rcnt[p] > 0:
                t = torch.empty(rcnt[p], dtype=torch.uint8)
                stash.append((p, t))
                reqs.append(dist.irecv(t, src=p))
        for p in range(world):
            if p == rank:
                if scnt[p] > 0:
                    C.memmove(recv + rdsp[p], send + sdsp[p], scnt[p])
                continue
            if scnt[p] > 0:
                reqs.append(dist.isend(_bytes_view(send + sdsp[p], scnt[p]).clone(), dst=p))
        for r in reqs:
            r.wait()
        for p, t in stash:
            C.memmove(recv + rdsp[p], t.numpy().ctypes.data, rcnt[p])

    return H.ALLREDUCE_FN(allreduce), H.ALLTOALLV_FN(alltoallv)


def init_host_transport(device: int = 0):
    """bicg_comm_init_host with gloo callbacks; torch.distributed must be initialised."""
    ar, a2a = callbacks()
    _keep.extend([ar, a2a])
    H.lib().bicg_comm_init_host(dist.get_rank(), dist.get_world_size(), ar, a2a, None, device)
    return ar, a2a


def init_rccl_transport(device: int):
    """RCCL communicator for the library, unique id broadcast over torch.distributed."""
    rank, world = dist.get_rank(), dist.get_world_size()
    ident = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_char * 128)()
        H.lib().bicg_comm_unique_id(buf)
        ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    dist.broadcast(ident, src=0)
    H.lib().bicg_comm_init_rccl(rank, world, bytes(ident.numpy().tobytes()), device)
