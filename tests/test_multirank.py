"""world_size-2 coverage of the N>1 path.

CPU (`-m "not gpu"`): the library's host-side halo planning driven over gloo, checked against the
oracle's distributed SpMV. GPU (`-m gpu`): two (and three) ranks share the single GPU of the box
through the host-staged transport and run the complete multi-rank solver -- the only piece the
8-GPU RCCL path does not share with it is the transport class (bicg_comm.cpp), whose RCCL calls are
exercised by bicg_comm_selftest_rccl."""
import glob
import os
import socket
import tempfile

import pytest
import torch.multiprocessing as mp

import mp_workers as W


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(fn, world, kind):
    with tempfile.TemporaryDirectory() as td:
        mp.start_processes(fn, args=(world, _free_port(), kind, td), nprocs=world, join=True, start_method="spawn")
        fails = glob.glob(os.path.join(td, "fail*"))
        assert not fails, open(fails[0]).read()
        assert len(glob.glob(os.path.join(td, "ok*"))) == world


@pytest.mark.parametrize("kind", ["offsets", "stencil", "ragged"])
@pytest.mark.parametrize("world", [2, 3])
def test_halo_plan_gloo(world, kind):
    _run(W.plan_worker, world, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world", [("offsets", 2), ("stencil", 2), ("ragged", 2), ("offsets", 3), ("ragged+nnz", 3)])
def test_multirank_solver_on_one_gpu(kind, world):
    _run(W.gpu_worker, world, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world", [("offsets+p2p", 2), ("stencil+p2p", 2), ("ragged+p2p", 3), ("ragged+nnz+p2p", 2), ("laplace+p2p", 4)])
def test_multirank_solver_peer_to_peer(kind, world):
    """Same checks with the peer-to-peer data path: the ranks map each other's mailboxes and halo
    rings through HIP IPC (here inside one GPU) and the kernels exchange LL words directly."""
    _run(W.gpu_worker, world, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["host", "p2p"])
def test_ranks_without_rows_take_part(kind):
    """6 rows over 8 ranks (reference src/matrix.c:295-298 gives ranks 6 and 7 nothing and runs): bicg_create no longer refuses"""
    _run(W.empty_rank_worker, 8, kind)


@pytest.mark.gpu
def test_rccl_single_rank_roundtrip():
    from mpi_bicgstab_amd import hipsolver as H
    assert H.lib().bicg_comm_selftest_rccl(0) == 0
