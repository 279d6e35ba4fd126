#!/usr/bin/env python3
"""bench.py -- ms/iteration and achieved HBM GB/s of the BiCGStab hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--method bicgstab]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE solver iteration (2 SpMV + the fused vector phases + the dot groups). Workload =
BASELINE.json configs[1]: plain BiCGStab on the Transport matrix on 1 GPU; Transport.mtx itself is
not available offline, so the matrix is the Transport-SHAPED synthetic of SURVEY.md section 8d
(n = 1 602 111, 15 diagonals, nnz = 23 921 209), symmetrically scaled over two decades so that the
W+K timed iterations are genuine unconverged iterations (the real Transport needs ~2700). Right-hand
side b = A*1, x0 = 0 (reference src/main.c:109-117). With N GPUs the SAME matrix is row-partitioned
exactly like the reference does (src/matrix.c:295-308): strong scaling; halo values and packed dot
sums are stored by the producing kernels straight into the other GPUs' memory over xGMI (HIP-IPC
mapped mailboxes, libbicgstab_hip.so's bicg_p2p.cpp; the IPC handles are exchanged over gloo at
set-up) once that path's self-test has passed on every rank, otherwise they go through an RCCL
communicator; torch.distributed (gloo) is only the bootstrap and the timing barrier.

Matrix and vectors are resident in HBM before the timed region. Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md (6290 measured copy)


def spmv_bytes(nnz, rows, halo=0):
    """val + col + ptr + x read once + y written once (+ halo in/out)."""
    return 12 * nnz + 4 * (rows + 1) + 8 * rows + 8 * rows + 16 * halo


ITER_VECTOR_BYTES_PER_ROW = {"bicgstab": 120, "ca_bicgstab": 184, "pipe_bicgstab": 192, "pipe_bicgstab_rr": 192}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--method", default="bicgstab", choices=list(ITER_VECTOR_BYTES_PER_ROW))
    ap.add_argument("--rows", dest="n", type=int, default=0, help="rows (default: 1602111)")
    ap.add_argument("--scale-decades", type=float, default=2.0)
    ap.add_argument("--workload", default="transport", choices=["transport", "laplace7"],
                    help="transport: BASELINE configs[1] (default). laplace7: 7-point Laplacian on an m^3 grid "
                         "(configs[3] is m = 512 over 8 GPUs = 64 planes of 512^2 per GPU)")
    ap.add_argument("--grid", dest="m", type=int, default=256, help="grid edge for --workload laplace7")
    ap.add_argument("--matrix", default=None, help="Matrix-Market file (coordinate real general) instead of the "
                    "synthetic; data/Transport.mtx is picked up automatically when it exists")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=100)
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "host", "host-p2p"],
                    help="auto (default): direct peer-to-peer stores over xGMI between the kernels (IPC handles "
                         "exchanged over gloo) when the library's self-test passes on every rank, otherwise an RCCL "
                         "communicator, one GPU per rank. rccl: RCCL collectives only. host: gloo-staged exchanges, ranks may share a GPU -- only for exercising "
                         "the multi-rank plumbing on a one-GPU box. host-p2p: the same with the peer-to-peer data path")
    ap.add_argument("--force-comm", action="store_true",
                    help="one rank only: run the multi-rank code path anyway (1-rank RCCL communicator) -- measures what "
                         "the transport adds to an iteration")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the JSON result: everything else this process or its libraries
    # print (gloo's connection report, RCCL's banner) is sent to stderr
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    # A hung collective must not hang the driver: after BENCH_WATCHDOG_S seconds every rank gives up.
    import threading

    def _give_up():
        if rank == 0:
            print(json.dumps({"metric": "ms/iteration", "value": None, "unit": "ms/iteration", "n_gpus": world,
                              "error": f"watchdog: no result after {watchdog_s} s (stage: {stage[0]})"}), file=result_out,
                  flush=True)
        os._exit(3)

    watchdog_s = int(os.environ.get("BENCH_WATCHDOG_S", "900"))
    stage = ["start"]
    dog = threading.Timer(watchdog_s, _give_up)
    dog.daemon = True
    dog.start()
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and a.gpus > 1:
            sys.exit(2)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the HIP path)")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)

    from mpi_bicgstab_amd import hipsolver as H
    from mpi_bicgstab_amd import synth
    L = H.lib()

    dist = None
    stage[0] = "communicator bootstrap"
    os.environ.setdefault("BICG_P2P_SOFT_FAIL", "1")     # a peer-to-peer time-out becomes a fallback, not an exit
    os.environ.setdefault("BICG_P2P_TIMEOUT_MS", "8000")   # ranks are aligned by barriers here: 8 s means a lost peer
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def comm_setup(use_p2p):
        """(Re)create the library's communicator; returns a description of the data path in use."""
        if world > 1:
            ident = torch.zeros(H_UNIQUE, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_char * H_UNIQUE)()
                L.bicg_comm_unique_id(buf)
                ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            dist.broadcast(ident, src=0)
            raw = bytes(ident.numpy().tobytes())
            from mpi_bicgstab_amd import dist_transport
            if a.transport == "auto" and use_p2p:
                # the peer-to-peer data path only needs a host-side exchange of IPC handles at set-up: bootstrap
                # it over gloo, so that RCCL is not even initialised unless the self-test fails somewhere
                dist_transport.init_host_transport(device)
                L.bicg_comm_enable_p2p()          # collective; every rank gets the same verdict
                if int(L.bicg_comm_p2p_active()):
                    mode = int(L.bicg_comm_p2p_active())
                    return mode, (f"peer-to-peer LL stores over xGMI (HIP IPC, {'uncached' if mode == 2 else 'device'} memory; "
                                  "bootstrap gloo)")
                L.bicg_comm_finalize()
                use_p2p = False
            if a.transport in ("auto", "rccl"):
                L.bicg_comm_init_rccl(rank, world, raw, device)
            else:
                dist_transport.init_host_transport(device)
            if use_p2p:
                L.bicg_comm_enable_p2p()      # collective; leaves the transport as it is when the self-test fails
        elif a.force_comm:
            os.environ["BICG_FORCE_COMM"] = "1"
            buf = (C.c_char * H_UNIQUE)()
            L.bicg_comm_unique_id(buf)
            L.bicg_comm_init_rccl(0, 1, buf.raw, device)
            if use_p2p:
                L.bicg_comm_enable_p2p()
        else:
            L.bicg_comm_init_single(device)
        mode = int(L.bicg_comm_p2p_active())
        base = "rccl" if a.transport in ("auto", "rccl") else "gloo-staged"
        if world == 1 and not a.force_comm:
            return 0, "none"
        if mode:
            return mode, (f"peer-to-peer LL stores over xGMI (HIP IPC, {'uncached' if mode == 2 else 'device'} memory; "
                          f"bootstrap {base})")
        return 0, base

    p2p_mode, transport_name = comm_setup(a.transport in ("auto", "host-p2p"))

    def barrier():
        if dist is not None:
            dist.barrier()

    note(f"communicator ready: {world} rank(s)")
    stage[0] = "matrix generation / upload"
    # ---- workload: this rank's row slab of the global matrix
    mtx = a.matrix or (os.path.join(ROOT, "data", "Transport.mtx") if a.workload == "transport" and not a.n and
                       os.path.exists(os.path.join(ROOT, "data", "Transport.mtx")) else None)
    if mtx:
        blocks = H.load_mtx_blocks(mtx, rank, world)
        n, nnz_global = int(blocks.info.rows), blocks.nnz_global
        counts, displs = synth.partition(n, world)
        lo, hi = int(displs[rank]), int(displs[rank] + counts[rank])
    elif a.workload == "laplace7":
        n = a.m ** 3
        nnz_global = synth.stencil7_nnz(a.m)
    else:
        n = a.n or synth.TRANSPORT_N
        nnz_global = synth.transport_nnz(n)
    if not mtx:
        counts, displs = synth.partition(n, world)
        lo, hi = int(displs[rank]), int(displs[rank] + counts[rank])
        if a.workload == "laplace7":
            slab = synth.stencil7(a.m, synth.LAPLACE_WEIGHTS, rows=(lo, hi))
        else:
            slab = synth.transport_like(n=n, rows=(lo, hi), scale_decades=a.scale_decades)
        diag, offd = synth.split_row_slab(slab, lo)
        blocks = H.HostBlocks(diag, offd if world > 1 else None, n, counts, displs)
    ones = np.ones(hi - lo)
    x0 = np.zeros(hi - lo)
    ctx = H.Context(blocks)
    plan = ctx.plan_info()
    b = ctx.spmv(ones)                       # b = A*1 (reference src/main.c:109-113), collective

    K, W = a.steps, a.warmup
    note(f"matrix resident: {plan}")
    stage[0] = "timed iterations"

    def timed_run(method, kernel_events):
        ctx.load(x0, b)
        ctx.run_begin(method, tol=0.0, max_iter=W + K, check_every=max(W, K, 1), krr=50, nrr=2,
                      time_kernels=1 if kernel_events else 0)
        if W:
            ctx.run_iterate(W)
        barrier(); ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.run_iterate(K)
        ctx.sync(); torch.cuda.synchronize(); barrier()
        dt = time.perf_counter() - t0
        res = ctx.run_end()
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        return dt, res

    def transport_check(res):
        """Did the exchanges deliver? The TRUE residual b - A x, recomputed with one more distributed
        SpMV, must agree with the recursive residual the iterations carried (plain BiCGStab keeps them
        within a small factor over a few hundred iterations). All ranks get the same answer."""
        x, r = ctx.fetch()
        tr = b - ctx.spmv(x)
        sums = torch.tensor([float(tr @ tr), float(r @ r), float(b @ b), 1.0 if ctx.comm_failed() else 0.0], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(sums)
        true_rel = float(np.sqrt(sums[0] / sums[2])) if sums[2] > 0 else float("nan")
        rec_rel = float(np.sqrt(sums[1] / sums[2])) if sums[2] > 0 else float("nan")
        ok = sums[3] == 0 and np.isfinite(true_rel) and np.isfinite(rec_rel) and true_rel <= 100.0 * rec_rel + 1e-12
        return bool(ok), true_rel

    # main timed region: exactly K iterations, no per-kernel instrumentation
    dt, res = timed_run(a.method, kernel_events=False)
    ok, true_relres = transport_check(res)
    if not ok and p2p_mode:
        # the peer-to-peer data path passed its self-test but not this check: measure with RCCL instead
        note(f"peer-to-peer data path failed the residual check (true relres {true_relres:.3e}); falling back")
        stage[0] = "fallback to the transport's collectives"
        failed_name = transport_name
        ctx.close()
        barrier()
        L.bicg_comm_finalize()
        p2p_mode, transport_name = comm_setup(False)
        transport_name += f" (fallback: '{failed_name}' failed the residual check)"
        ctx = H.Context(blocks)
        plan = ctx.plan_info()
        b = ctx.spmv(ones)
        dt, res = timed_run(a.method, kernel_events=False)
        ok, true_relres = transport_check(res)
    ms_step = 1e3 * dt / K
    relres = float(np.sqrt(res.dot_r / res.dot_zero)) if res.dot_zero > 0 else float("nan")
    genuine = res.iterations == W + K and np.isfinite(relres) and ok

    note(f"{a.method}: {ms_step:.4f} ms/iteration")
    stage[0] = "roofline / variant legs"
    # roofline leg: the same K iterations, every SpMV kernel launched with its own start/stop HIP events
    # (hipExtLaunchKernelGGL on the library's compute stream): kernel durations, no launch gaps
    dt_ev, res_ev = timed_run(a.method, kernel_events=True)
    spmv_ms = res_ev.spmv_ms_total / max(res_ev.spmv_launches, 1)
    b_spmv = spmv_bytes(plan["nnz_diag"] + plan["nnz_offd"], plan["rows"], plan["halo"])
    achieved = b_spmv / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_spmv.json")
    if os.path.exists(pmc) and a.workload == "transport" and not a.n and world == 1 and not mtx:
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    variants = {}
    if not a.no_variants:
        for m in ITER_VECTOR_BYTES_PER_ROW:
            if m == a.method:
                variants[m] = ms_step
                continue
            dtv, resv = timed_run(m, kernel_events=False)
            variants[m] = 1e3 * dtv / K
    if not a.no_variants and a.workload == "transport":
        # BASELINE.json configs[4] family: 16 shifts, seed 7, sigma_j = (j+1) 0.01/16 (reference
        # src/main_shifted.c:99 pattern); 2 SpMV + one batched update over all shifts per iteration
        nsh, seed = 16, 7
        sigma = (np.arange(nsh) + 1.0) * 0.01 / nsh
        ks = min(K, 100)
        for which in ("shifted_lopbicgstab", "shifted_pipe_lopbicgstab"):
            rs = ctx.solve_shifted(b + sigma[seed] * ones, sigma, seed, tol=0.0, max_iter=ks, check_every=ks, which=which)
            variants[f"{which}_{nsh}shifts"] = 1e3 * rs["result"].seconds / max(rs["k"], 1)
    spmv_alone_ms = ctx.spmv_bench(200)

    cpu = None
    cpu_all = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload == "transport" and not mtx:
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--n", str(n),
                                  "--scale-decades", str(a.scale_decades), "--iters", str(a.cpu_iters),
                                  "--method", a.method], capture_output=True, text=True, timeout=900)
            cpu = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # the baseline is reported, never required
            cpu = {"error": repr(e)}
        # SURVEY.md section 8d config 1 also asks for the reference on the host's cores. More ranks are not
        # faster for it: every rank gathers the WHOLE vector per SpMV (src/matrix.c:432), and on the GPU box's
        # 2 x EPYC 9575F it is fastest at 8 ranks (ms/iteration at 4/8/16/32/64 ranks: 15.5 / 11.7 / 13.1 /
        # 53.4 / 135, profiles/r01/cpu_reference_scaling.txt) -- so 8 ranks is what is timed here
        try:
            ranks = max(2, min(8, os.cpu_count() or 2))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--n", str(n),
                                  "--scale-decades", str(a.scale_decades), "--iters", str(a.cpu_iters),
                                  "--method", a.method, "--ranks", str(ranks)], capture_output=True, text=True, timeout=600)
            cpu_all = json.loads(out.stdout.strip().splitlines()[-1])
            cpu_all["note"] = "rank count at which the reference is fastest on this host; see profiles/r01/cpu_reference_scaling.txt"
        except Exception as e:
            cpu_all = {"error": repr(e)}

    if rank == 0:
        iter_bytes = 2 * spmv_bytes(nnz_global, n) + ITER_VECTOR_BYTES_PER_ROW[a.method] * n
        line = {
            "metric": f"ms/iteration, {a.method}, " + ("Transport-shaped CSR" if a.workload == "transport" else f"7-point Laplacian {a.m}^3")
                      + f" ({n} rows, {nnz_global} nnz), strong scaling over GPUs",
            "value": ms_step, "unit": "ms/iteration", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_step, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": ("synthetic" if not mtx else "file:" + os.path.basename(mtx)),
            "config": {"workload": ("BASELINE.json configs[1]: plain BiCGStab, Transport-shaped synthetic "
                                    "(Transport.mtx unavailable offline), b = A*1, x0 = 0") if a.workload == "transport" else
                                   f"BASELINE.json configs[3] family: 7-point 3-D Laplacian {a.m}^3 generated in memory, b = A*1, x0 = 0",
                       "rows": n, "nnz": nnz_global, "scale_decades": a.scale_decades, "method": a.method,
                       "partition": f"row blocks over {world} GPU(s), reference src/matrix.c:295-308",
                       "transport": transport_name,
                       "iterations_genuine": bool(genuine), "relres_after_timed_region": relres,
                       "true_relres_after_timed_region": true_relres},
            "hbm_gbps_iteration": iter_bytes / (ms_step * 1e-3) / 1e9,
            "iteration_algorithmic_bytes": iter_bytes,
            "roofline": {"kernel": "k_spmv_sell (sliced-ELL SpMV with fused dot epilogue), rank 0 share", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": b_spmv,
                         "avg_launch_ms": spmv_ms, "launches_timed": res_ev.spmv_launches,
                         "ms_per_step_with_events": 1e3 * dt_ev / K,
                         "back_to_back_spmv_ms": spmv_alone_ms,
                         "frac_of_measured_copy_6290": achieved / 6290.0},
            "cpu_baseline": cpu,
            "cpu_baseline_multicore": cpu_all,
            "variants_ms_per_iteration": variants,
        }
        print(json.dumps(line), file=result_out, flush=True)

    dog.cancel()
    ctx.close()
    if dist is not None:
        dist.barrier()
        L.bicg_comm_finalize()
        dist.destroy_process_group()


H_UNIQUE = 128

if __name__ == "__main__":
    main()
