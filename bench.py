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
communicator, and if that cannot be created either, through gloo-staged exchanges;
torch.distributed (gloo) is otherwise only the bootstrap and the timing barrier.

Matrix and vectors are resident in HBM before the timed region. Rank 0 prints ONE JSON line; besides
the headline it carries the other workloads north_star names (`extras`: synthetic banded CSR at
half-bandwidth 8 / 64 / 512 at every GPU count; on one GPU also an irregular FEM-like matrix and the
256^3 Laplacian = one GPU's share of configs[3] under CA-BiCGStab) and the roofline fraction of every
solver variant on the headline matrix (`variant_rooflines`).
"""
from __future__ import annotations

import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

# the host driver of this pool only supports dmabuf IPC: without this the HIP-IPC mapping of the peer-to-peer
# mailboxes (and RCCL's own IPC) fails with "hipIpcGetMemHandle: invalid argument". Must be set before the HIP
# runtime initialises; normally exported already.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md; the box's own STREAM rates are measured (stream leg)
H_UNIQUE = 128


def spmv_bytes(nnz, rows, halo=0):
    """val + col + ptr + x read once + y written once (+ halo in/out)."""
    return 12 * nnz + 4 * (rows + 1) + 8 * rows + 8 * rows + 16 * halo


def csr_bytes(nnz, rows):
    """the matrix arrays of section 8d's product: val + col + ptr"""
    return 12 * nnz + 4 * (rows + 1)


# fused-minimum vector bytes per row and iteration (SURVEY.md section 8d "Algorithmic bytes, iteration")
ITER_VECTOR_BYTES_PER_ROW = {"bicgstab": 120, "ca_bicgstab": 184, "pipe_bicgstab": 192, "pipe_bicgstab_rr": 192}


def iteration_bytes(method, nnz, rows):
    return 2 * spmv_bytes(nnz, rows) + ITER_VECTOR_BYTES_PER_ROW[method] * rows


def shifted_iteration_bytes(nshift, nnz, rows, pipelined):
    """config 5: 2 SpMV + the batched multi-shift update (32 n per non-seed shift,
    p_j and x_j read and written once) + the seed's own vectors (plain 120 n, pipelined 192 n)"""
    return 2 * spmv_bytes(nnz, rows) + (nshift - 1) * 32 * rows + (192 if pipelined else 120) * rows


MALL_BYTES = 256 * 1048576      # Infinity Cache
NVEC = {"bicgstab": 6, "ca_bicgstab": 8, "pipe_bicgstab": 10, "pipe_bicgstab_rr": 11}


def roof(alg_bytes, seconds, csr_bytes, spmv_stream_bytes, nspmv, rows, nvec, flags, stream, world=1, spmv_only=False):
    """Which resource bounds a leg, and `frac` against THAT bound only. `frac` is claimed on the bytes the STORED FORMAT has to
    move (`format_bytes`: the matrix arrays one product streams -- bicg_spmv_matrix_bytes: values + whatever index the layout
    keeps, 16-bit offsets or none at all in uniform slices -- plus the vector traffic of SURVEY.md section 8d), never on the CSR
    figure of section 8d (`algorithmic_bytes` / `gbps`, kept beside it): a layout that stores no column index moves fewer bytes
    than 12 per non-zero, and a fraction formed with the CSR figure would exceed 1.
      latency: a rank the persistent kernels hold in LDS / registers (one launch per chunk of iterations): its time is a chain
               of dependent hand-offs between workgroups, not bytes -- no bandwidth fraction is claimed (frac null);
      mall:    matrix + vectors fit the 256 MiB Infinity Cache: peak = this GPU's read rate out of the Infinity Cache, measured
               in this run (bicg_stream_bench on a 96 MiB array); frac null when that probe did not run;
      hbm:     everything else: peak = 8 TB/s (MI355X_MICROARCH.md)."""
    # csr_bytes: 12 nnz + 4 (n + 1) of the whole matrix (section 8d); spmv_stream_bytes: what this rank's layout streams per product
    fmt = alg_bytes - nspmv * max(0, csr_bytes - world * spmv_stream_bytes)
    fgb = fmt / seconds / 1e9 / world
    ws = spmv_stream_bytes + 8 * rows * nvec
    out = dict(format_bytes=int(fmt), format_gbps=fgb * world)
    if not spmv_only and "persist" in flags and rows <= 250_000:
        return dict(out, bound="latency", peak=None, frac=None)
    if ws <= MALL_BYTES:
        peak = stream.get("mall_read8") if isinstance(stream, dict) else None
        return dict(out, bound="mall", peak=peak, frac=(fgb / peak) if peak else None)
    return dict(out, bound="hbm", peak=HBM_PEAK_GBS, frac=fgb / HBM_PEAK_GBS)


def _r(v, nd=6):
    """numbers of the compact line: 6 significant digits are more than any clock here resolves"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float(f"{v:.{nd}g}") if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


COMPACT_LIMIT = 8192      # bytes; the driver's capture parsed a 16 KB line and lost a 27 KB one (round 5) -- stay well below
ROOF_KEYS = ("kernel", "bound", "peak", "unit", "achieved", "frac", "frac_basis", "survey_8d_frac", "algorithmic_bytes_per_launch",
             "avg_launch_ms", "traffic", "frac_of_measured_copy")


def compact(full):
    """The ONE line rank 0 prints on stdout, formed from the complete record (which goes to bench_full.json and stderr).
    Contract keys + roofline + cpu_baseline + one {ms_per_iteration, bound, frac} per extra leg; long texts are cut."""
    def cut(s, n):
        return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"

    def roofline(r, kernel_chars=96):
        if not r:
            return None
        out = {k: r.get(k) for k in ROOF_KEYS}
        out["kernel"] = cut(out["kernel"], kernel_chars)
        out["frac_basis"] = cut(out["frac_basis"], 96)
        if out["algorithmic_bytes_per_launch"] is None:
            out["algorithmic_bytes_per_launch"] = r.get("format_bytes_per_launch")
        sd = r.get("structure_dependence")
        if sd:
            out["structure_dependence"] = {"ms_per_iteration": sd.get("ms_per_iteration"), "frac": sd.get("frac_of_format_bytes", sd.get("frac"))}
        if isinstance(r.get("ms_per_iteration"), dict):
            out["ms_per_iteration"] = r["ms_per_iteration"]
        for k in ("numbering", "product_kernels", "survey_8d_frac_of_measured_copy"):
            if k in r:
                out[k] = r[k]
        if isinstance(r.get("other_numberings"), dict):
            out["other_numberings"] = {kind: {k: cut(u.get(k), 160) for k in ("product_kernels", "avg_launch_ms", "frac", "survey_8d_frac",
                                                                              "survey_8d_frac_of_measured_copy", "ms_per_iteration", "bottleneck") if k in u}
                                       for kind, u in r["other_numberings"].items()}
        st = r.get("stream_measured_gbps")
        if isinstance(st, dict):
            out["stream_measured_gbps"] = {k: v for k, v in st.items() if k != "note"}
        return out

    def leg(e):
        """an extra workload: one {ms_per_iteration, bound, frac} per method it ran"""
        if not isinstance(e, dict):
            return e
        if "error" in e:
            return {"error": cut(e["error"], 120)}
        out = {}
        for k, v in e.items():
            if isinstance(v, dict) and "ms_per_iteration" in v:
                out[k] = {"ms_per_iteration": v["ms_per_iteration"], "bound": v.get("bound"), "frac": v.get("frac")}
                if v.get("iterations_genuine") is False:
                    out[k]["iterations_genuine"] = False
            elif isinstance(v, dict) and "ms" in v and "bound" in v:       # spmv_back_to_back / spmm legs
                out[k] = {"ms": v["ms"], "bound": v.get("bound"), "frac": v.get("frac")}
        if "ms_per_iteration" in e:
            out.update(ms_per_iteration=e["ms_per_iteration"], bound=e.get("bound"), frac=e.get("frac"))
        elif "ms" in e and "bound" in e:        # a leg that is one kernel (the SpMM)
            out.update(ms=e["ms"], bound=e.get("bound"), frac=e.get("frac"))
        return out

    cfg = full.get("config", {})
    cpu, cpu_all = full.get("cpu_baseline"), full.get("cpu_baseline_multicore")
    cm = full.get("comm") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["metric"] = cut(line["metric"], 140)
    line["config"] = {k: cfg.get(k) for k in ("workload", "rows", "nnz", "method", "transport", "iterations_genuine", "halo")}
    line["config"]["workload"] = cut(line["config"]["workload"], 140)
    line["config"]["transport"] = cut(line["config"]["transport"], 120)
    line["roofline"] = roofline(full.get("roofline"))
    line["roofline_unstructured"] = roofline(full.get("roofline_unstructured"))
    if isinstance(cpu, dict):
        line["cpu_baseline"] = {k: cut(cpu.get(k), 140) for k in ("value", "unit", "cores", "kind", "sample", "flags", "error") if k in cpu}
    else:
        line["cpu_baseline"] = cpu
    if isinstance(cpu_all, dict):
        line["cpu_baseline_multicore"] = {k: cut(cpu_all.get(k), 120) for k in ("value", "unit", "cores", "kind", "error") if k in cpu_all}
    else:
        line["cpu_baseline_multicore"] = cpu_all
    rl = cm.get("rccl_leg")
    if isinstance(rl, dict):
        rl = {k: cut(v, 160) for k, v in rl.items() if k in ("ms_per_iteration", "unavailable", "rccl_nranks", "iterations_genuine", "note", "attempt")}
    line["comm"] = {"world": cm.get("world"), "p2p_selftest": cm.get("p2p_selftest"), "rccl_nranks": cm.get("rccl_nranks"),
                    "transport_used": cut(cm.get("transport_used"), 120), "fallback_reason": cut(cm.get("fallback_reason"), 160),
                    "wait_us": cm.get("wait_us"), "rccl_leg": rl}
    line["variants_ms_per_iteration"] = full.get("variants_ms_per_iteration")
    line["variants_frac"] = {k: v.get("frac") for k, v in (full.get("variant_rooflines") or {}).items()}
    line["extras"] = {k: leg(v) for k, v in (full.get("extras") or {}).items()}
    line["full_record"] = full.get("full_record")
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= COMPACT_LIMIT:        # never expected: drop the optional blocks rather than lose the whole line
        for k in ("extras", "variants_frac", "variants_ms_per_iteration"):
            line[k] = None
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < COMPACT_LIMIT:
                break
    return text


def measure_traffic(argv_inner, note):
    """HBM bytes per SpMV launch of THIS run's kernel, from rocprofv3 PMC counters collected now: two
    separate passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; the TCC block cannot hold both) over a short
    inner run of this script. Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section):
    both counters are KiB, FETCH_SIZE tallies 128-byte requests as 64 -> x2. The x2 is re-checked in
    the same pass on k_vec<FPlainQ>, whose read bytes are known exactly (2 vectors of n doubles);
    the SpMV issues 8-byte loads, for which the guide does not state the factor -- the figure is the
    measured counters under the factor that reproduces the element-wise kernel."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="bicg_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [rocprof, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "bench.py")] + argv_inner
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                note(f"traffic: no counter file from the {ctr} pass (rc {r.returncode})")
                return None
            acc = {}
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") != ctr:
                    continue
                acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            out[ctr] = acc
    except Exception as e:  # reported, never required
        note(f"traffic: {e!r}")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def per_launch(ctr, key):
        vals = [v for name, lst in out[ctr].items() if key(name) for v in lst]
        return (1024.0 * sum(vals) / len(vals), len(vals)) if vals else (None, 0)

    # the in-solver SpMV with its dot epilogue (sliced-ELL kernel, or the ragged-rows product of bicg_jagw.hip)
    dots = lambda nm: any(k in nm for k in ("k_spmv_sell<1", "k_spmv_sell<2", "k_spmv_jagw<1", "k_spmv_jagw<2", "k_spmv_jagd<1", "k_spmv_jagd<2", "k_spmv_jagl<1", "k_spmv_jagl<2"))
    fetch, nf = per_launch("FETCH_SIZE", dots)
    write, nw = per_launch("WRITE_SIZE", dots)
    qf, _ = per_launch("FETCH_SIZE", lambda nm: "FPlainQ" in nm)
    if fetch is None or write is None:
        return None
    return dict(fetch_counter_bytes=fetch, write_counter_bytes=write, launches=min(nf, nw), fplainq_fetch_counter_bytes=qf)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--regions", type=int, default=3, help="timed regions of exactly --steps iterations each for the headline; "
                    "`value` is their median, every region goes into the line with its three clocks (headline_regions)")
    ap.add_argument("--method", default="bicgstab", choices=list(ITER_VECTOR_BYTES_PER_ROW))
    ap.add_argument("--rows", dest="n", type=int, default=0, help="rows (default: 1602111; banded: ~24 M non-zeros)")
    ap.add_argument("--scale-decades", type=float, default=2.0)
    ap.add_argument("--numbering", default="rcm", choices=["generator", "rcm", "random"], help="--workload mesh: the node numbering")
    ap.add_argument("--workload", default="transport", choices=["transport", "laplace7", "banded", "fem_like", "mesh"],
                    help="transport: BASELINE configs[1] (default, the headline). banded: dense band, --half-bandwidth b "
                         "(SURVEY 8d synthetic input ii). fem_like: irregular rows (3..27 per row) on the Transport node "
                         "numbering. laplace7: 7-point Laplacian on an m^3 grid (configs[3] is m = 512 over 8 GPUs = 64 "
                         "planes of 512^2 per GPU). mesh: the unstructured FEM matrix of mpi_bicgstab_amd.mesh (P1 elements on a "
                         "Delaunay tetrahedralisation of 117^3 jittered points: 1 601 613 rows, 26.0 M non-zeros, symmetric pattern, "
                         "unsymmetric values -- the stand-in for Transport.mtx) in the numbering --numbering")
    ap.add_argument("--half-bandwidth", type=int, default=64)
    ap.add_argument("--grid", dest="m", type=int, default=256, help="grid edge for --workload laplace7")
    ap.add_argument("--matrix", default=None, help="Matrix-Market file (coordinate real general) instead of the "
                    "synthetic; data/Transport.mtx is picked up automatically when it exists")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the banded / FEM-like / Laplacian legs")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic = null)")
    ap.add_argument("--no-laplace512", action="store_true", help="skip the 512^3 Laplacian leg (needs ~45 GB of HBM and ~15 s)")
    ap.add_argument("--no-stream", action="store_true", help="skip the STREAM probes (roofline.stream_measured_gbps = null)")
    ap.add_argument("--no-rccl-leg", action="store_true", help="N > 1: do not time the headline a second time over RCCL collectives")
    ap.add_argument("--cpu-iters", type=int, default=100)
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "host", "host-p2p"],
                    help="auto (default): direct peer-to-peer stores over xGMI between the kernels (IPC handles "
                         "exchanged over gloo) when the library's self-test passes on every rank, otherwise an RCCL "
                         "communicator, one GPU per rank. rccl: RCCL collectives only. host: gloo-staged exchanges, ranks may share a GPU -- only for exercising "
                         "the multi-rank plumbing on a one-GPU box. host-p2p: the same with the peer-to-peer data path")
    ap.add_argument("--force-comm", action="store_true",
                    help="one rank only: run the multi-rank code path anyway (1-rank RCCL communicator) -- measures what "
                         "the transport adds to an iteration")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)   # the run rocprofv3 wraps for roofline.traffic
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, the command the driver uses)
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"bench.py: --gpus {a.gpus} without a launcher: re-executing under torch.distributed.run", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

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
            msg = str(msg)       # stderr shares the driver's bounded capture with the result line: notes stay short
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg if len(msg) <= 400 else msg[:399] + '~'}", file=sys.stderr, flush=True)

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
    os.environ.setdefault("BICG_P2P_SOFT_FAIL", "1")      # a peer-to-peer time-out becomes a fallback, not an exit
    os.environ.setdefault("BICG_COMM_SOFT_FAIL", "1")     # so does an RCCL communicator that cannot be created
    os.environ.setdefault("BICG_P2P_TIMEOUT_MS", "8000")  # ranks are aligned by barriers here: 8 s means a lost peer
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def everyone(ok):
        """True when `ok` holds on every rank (collective over gloo)"""
        if dist is None:
            return bool(ok)
        t = torch.tensor([0 if ok else 1], dtype=torch.int32)
        dist.all_reduce(t)
        return int(t[0]) == 0

    comm_info = {"world": world, "p2p_selftest": None, "rccl_nranks": None, "transport_used": None, "fallback_reason": None, "wait_us": None}

    def p2p_label(mode, base):
        """what the peer-to-peer data path crossed: xGMI links only when every rank has a GPU of its own"""
        mem = "uncached" if mode == 2 else "device"
        if world == 1:      # --force-comm: one rank drives the whole path, its mailboxes are its own memory
            return (f"peer-to-peer LL stores of ONE rank into its own mailboxes ({mem} memory; no peer, no xGMI link involved; "
                    f"bootstrap {base})")
        if world > torch.cuda.device_count():
            return (f"peer-to-peer LL stores between processes SHARING one device (HIP IPC, {mem} memory; no xGMI link involved; "
                    f"bootstrap {base})")
        return f"peer-to-peer LL stores over xGMI (HIP IPC, {mem} memory; bootstrap {base})"

    def comm_setup(use_p2p, allow_rccl=True, want="auto"):
        """(Re)create the library's communicator; returns a description of the data path in use."""
        from mpi_bicgstab_amd import dist_transport
        transport = a.transport if want == "auto" else want
        if world > 1:
            ident = torch.zeros(H_UNIQUE, dtype=torch.uint8)
            if transport == "auto" and use_p2p:
                # the peer-to-peer data path only needs a host-side exchange of IPC handles at set-up: bootstrap
                # it over gloo, so that RCCL is not even initialised unless the self-test fails somewhere
                dist_transport.init_host_transport(device)
                rc_p2p = int(L.bicg_comm_enable_p2p())          # collective; every rank gets the same verdict
                comm_info["p2p_selftest"] = "passed" if rc_p2p == 0 and int(L.bicg_comm_p2p_active()) else f"failed (code {rc_p2p})"
                if int(L.bicg_comm_p2p_active()):
                    mode = int(L.bicg_comm_p2p_active())
                    return mode, p2p_label(mode, "gloo")
                L.bicg_comm_finalize()
                use_p2p = False
            base = "gloo-staged"
            if transport in ("auto", "rccl") and allow_rccl:
                if rank == 0:
                    buf = (C.c_char * H_UNIQUE)()
                    L.bicg_comm_unique_id(buf)
                    ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
                dist.broadcast(ident, src=0)
                rc = L.bicg_comm_init_rccl(rank, world, bytes(ident.numpy().tobytes()), device)
                if rc != 0:
                    ebuf = C.create_string_buffer(512)
                    L.bicg_comm_last_error(ebuf, 512)
                    comm_info["rccl_error"] = ebuf.value.decode(errors="replace")
                if everyone(rc == 0):
                    base = "rccl"
                    comm_info["rccl_nranks"] = int(L.bicg_comm_size())
                else:           # some rank could not join: every rank drops RCCL, exchanges are staged through gloo
                    note("RCCL communicator could not be created on every rank: falling back to gloo-staged exchanges")
                    L.bicg_comm_finalize()
                    dist_transport.init_host_transport(device)
                    base = "gloo-staged (fallback: RCCL communicator could not be created)"
                    comm_info["rccl_nranks"] = 0
            else:
                dist_transport.init_host_transport(device)
            if use_p2p:
                rc_p2p = int(L.bicg_comm_enable_p2p())      # collective; leaves the transport as it is when the self-test fails
                comm_info["p2p_selftest"] = "passed" if rc_p2p == 0 and int(L.bicg_comm_p2p_active()) else f"failed (code {rc_p2p})"
        elif a.force_comm:
            H.switches(force_comm=1)
            buf = (C.c_char * H_UNIQUE)()
            L.bicg_comm_unique_id(buf)
            L.bicg_comm_init_rccl(0, 1, buf.raw, device)
            base = "rccl"
            if use_p2p:
                L.bicg_comm_enable_p2p()
        else:
            L.bicg_comm_init_single(device)
            return 0, "none"
        mode = int(L.bicg_comm_p2p_active())
        if mode:
            return mode, p2p_label(mode, base)
        return 0, base

    p2p_mode, transport_name = comm_setup(a.transport in ("auto", "host-p2p"))

    # ------------------------------------------------------------------ this GPU's own STREAM rates
    stream = None
    if not a.no_stream and not a.inner:
        stage[0] = "STREAM probes"
        shared = world > torch.cuda.device_count()      # ranks sharing a GPU (tests): rank 0 measures, the others wait
        try:
            if dist is not None:
                dist.barrier()
            if not shared or rank == 0:
                stream = {k: max(H.stream_bench(k, 1 << 30, 10) for _ in range(2)) for k in ("copy", "triad", "read8", "read16")}
                # the same read loop on an array that fits the Infinity Cache: the denominator of the "mall"-bound legs
                stream["mall_read8"] = max(H.stream_bench("read8", 96 << 20, 40) for _ in range(2))
            if dist is not None:
                dist.barrier()
            if stream is None:
                raise RuntimeError("measured by rank 0 only (ranks share the device)")
            stream["note"] = ("GB/s of bytes read + written, 1 GiB per array (4 x the Infinity Cache), fastest of {grid-stride, one "
                              "workgroup per 16 KiB tile} x {ordinary, non-temporal} accesses; mall_read8: the read loop on a 96 MiB "
                              "array (served by the Infinity Cache); libbicgstab_hip.so bicg_stream_bench")
            note("STREAM on this GPU: " + ", ".join(f"{k} {v:.0f} GB/s" for k, v in stream.items() if k != "note"))
        except Exception as e:  # reported, never required
            stream = {"error": repr(e)}

    def barrier():
        if dist is not None:
            dist.barrier()

    def base_is_rccl(name):
        return name == "rccl" or "bootstrap rccl" in name

    note(f"communicator ready: {world} rank(s), data path: {transport_name}")
    K, W = a.steps, a.warmup

    # ------------------------------------------------------------------ workloads: this rank's row slab
    mesh_cache = os.path.join(tempfile.gettempdir(), "bicg_mesh_cache")

    def mesh_slab(kind, lo_hi):
        """this rank's rows of the unstructured FEM matrix; rank 0 generates (or finds) the assembled matrix first, the other ranks
        of the host read its files"""
        from mpi_bicgstab_amd import mesh
        os.makedirs(mesh_cache, exist_ok=True)
        if rank == 0:
            slab = mesh.fem_unstructured(117, kind, a.scale_decades, rows=lo_hi, cache_dir=mesh_cache)
        barrier()
        if rank != 0:
            slab = mesh.fem_unstructured(117, kind, a.scale_decades, rows=lo_hi, cache_dir=mesh_cache)
        return slab

    def build(workload, n=0, half_bw=64, m=256, matrix=None, numbering="rcm"):
        """-> dict(name, rows, nnz, blocks, lo, hi, note): this rank's blocks of the global matrix"""
        mtx = matrix or (os.path.join(ROOT, "data", "Transport.mtx") if workload == "transport" and not n and
                         os.path.exists(os.path.join(ROOT, "data", "Transport.mtx")) else None)
        if mtx:
            blocks = H.load_mtx_blocks(mtx, rank, world)
            rows, nnz = int(blocks.info.rows), blocks.nnz_global
            counts, displs = synth.partition(rows, world)
            lo, hi = int(displs[rank]), int(displs[rank] + counts[rank])
            return dict(name="file:" + os.path.basename(mtx), rows=rows, nnz=nnz, blocks=blocks, lo=lo, hi=hi, data="file:" + os.path.basename(mtx),
                        desc=f"{os.path.basename(mtx)} through the library's block loader, b = A*1, x0 = 0")
        if workload == "laplace7":
            rows, nnz = m ** 3, synth.stencil7_nnz(m)
            desc = f"BASELINE.json configs[3] family: 7-point 3-D Laplacian {m}^3 generated in memory, b = A*1, x0 = 0"
            gen = lambda lo, hi: synth.stencil7(m, synth.LAPLACE_WEIGHTS, rows=(lo, hi))
        elif workload == "banded":
            rows = n or synth.banded_rows_for(24_000_000, half_bw)
            nnz = synth.banded_nnz(rows, half_bw)
            desc = (f"synthetic banded CSR (SURVEY.md 8d input ii): half-bandwidth {half_bw}, {rows} rows, value law of the "
                    f"Transport-shaped synthetic scaled over {a.scale_decades} decades, b = A*1, x0 = 0")
            gen = lambda lo, hi: synth.banded(rows, half_bw, rows=(lo, hi), scale_decades=a.scale_decades)
        elif workload == "mesh":
            rows = 117 ** 3
            nnz = None
            desc = (f"unstructured FEM matrix (mesh.py: P1 convection-diffusion on a Delaunay tetrahedralisation of 117^3 jittered points, "
                    f"symmetric pattern, unsymmetric values), numbering '{numbering}', values scaled over {a.scale_decades} decades, b = A*1, x0 = 0")
            gen = lambda lo, hi: mesh_slab(numbering, (lo, hi))
        elif workload == "fem_like":
            rows = n or synth.TRANSPORT_N
            nnz = None
            desc = ("irregular FEM-like CSR: 27-point stencil on the Transport node numbering, every off-diagonal kept with "
                    f"probability 0.55 (row lengths 6..27), values scaled over {a.scale_decades} decades, b = A*1, x0 = 0")
            gen = lambda lo, hi: synth.fem_like(rows, rows=(lo, hi), scale_decades=a.scale_decades)
        else:
            rows = n or synth.TRANSPORT_N
            nnz = synth.transport_nnz(rows)
            desc = ("BASELINE.json configs[1]: plain BiCGStab, Transport-shaped synthetic (Transport.mtx unavailable offline), "
                    "b = A*1, x0 = 0")
            gen = lambda lo, hi: synth.transport_like(n=rows, rows=(lo, hi), scale_decades=a.scale_decades)
        counts, displs = synth.partition(rows, world)
        lo, hi = int(displs[rank]), int(displs[rank] + counts[rank])
        slab = gen(lo, hi)
        if nnz is None:
            t = torch.tensor([slab.nnz], dtype=torch.int64)
            if dist is not None:
                dist.all_reduce(t)
            nnz = int(t[0])
        diag, offd = synth.split_row_slab(slab, lo)
        blocks = H.HostBlocks(diag, offd if world > 1 else None, rows, counts, displs)
        return dict(name=workload, rows=rows, nnz=nnz, blocks=blocks, lo=lo, hi=hi, data="synthetic", desc=desc)

    regions_ms = {}            # every timed region of the variant / extra legs, per leg and method (ms per region)
    region_clocks = {}         # ... and its three clocks (host region, device events, host enqueue time)

    class Leg:
        """one resident matrix: context, right-hand side, timed solves"""

        def __init__(self, wl):
            self.wl = wl
            t_setup = time.perf_counter()
            self.ctx = H.Context(wl["blocks"])      # bicg_create: plan on the host's threads + upload + code objects
            self.setup_s = time.perf_counter() - t_setup
            self.plan = self.ctx.plan_info()
            self.ones = np.ones(wl["hi"] - wl["lo"])
            self.x0 = np.zeros(wl["hi"] - wl["lo"])
            self.b = self.ctx.spmv(self.ones)         # b = A*1 (reference src/main.c:109-113), collective

        def close(self):
            self.ctx.close()

        def best(self, method, steps=None, warm=None, tries=2):
            """the variant / extra legs: the faster of two timed regions; BOTH go into the line (`timed_regions_ms`), so a
            region that stalled is on record. (Rounds 2-3 saw one region per few runs take 10-80 ms longer, never twice in a
            row. What was found in round 4: a translation unit's code object is loaded at the first look-up of one of its
            kernels -- 4-10 ms, now paid inside bicg_create (preload_kernels) -- and nothing longer than that could be
            provoked outside this script: profiles/r04/preload_check.txt. The headline `value` is a single region of
            exactly K iterations.)"""
            runs = []
            for _ in range(tries):
                runs.append(self.timed(method, steps=steps, warm=warm))
                key = f'{self.wl.get("desc", "headline")} / {method}'
                regions_ms.setdefault(key, []).append(round(1e3 * runs[-1][0], 4))
                region_clocks.setdefault(key, []).append(self.last_clocks)
            return min(runs, key=lambda r: r[0])

        def timed(self, method, kernel_events=False, steps=None, warm=None):
            k, w = steps or K, W if warm is None else warm
            ctx = self.ctx
            ctx.load(self.x0, self.b)
            ctx.run_begin(method, tol=0.0, max_iter=w + k, check_every=max(w, k, 1), krr=50, nrr=2,
                          time_kernels=1 if kernel_events else 0)
            if w:
                ctx.run_iterate(w)
            barrier(); ctx.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, clocks = ctx.run_iterate_timed(k)      # the K iterations; device events + enqueue time + wall time taken inside the library
            ctx.sync(); torch.cuda.synchronize(); barrier()
            dt = time.perf_counter() - t0
            res = ctx.run_end()
            if dist is not None:
                t = torch.tensor([dt], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t[0])
            # a region that takes long on the host's clock but not between the device's events was held up outside the kernels
            self.last_clocks = dict(region_ms=round(1e3 * dt, 4), device_ms=round(clocks["device_ms"], 4),
                                    enqueue_ms=round(clocks["enqueue_ms"], 4), library_wall_ms=round(clocks["wall_ms"], 4))
            return dt, res

        def check(self):
            """Did the exchanges deliver? The TRUE residual b - A x, recomputed with one more distributed
            SpMV, must agree with the recursive residual the iterations carried (plain BiCGStab keeps them
            within a small factor over a few hundred iterations). All ranks get the same answer."""
            x, r = self.ctx.fetch()
            tr = self.b - self.ctx.spmv(x)
            sums = torch.tensor([float(tr @ tr), float(r @ r), float(self.b @ self.b), 1.0 if self.ctx.comm_failed() else 0.0],
                                dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(sums)
            s = [float(v) for v in sums]
            true_rel = float(np.sqrt(s[0] / s[2])) if s[2] > 0 else float("nan")
            rec_rel = float(np.sqrt(s[1] / s[2])) if s[2] > 0 else float("nan")
            ok = s[3] == 0 and np.isfinite(true_rel) and np.isfinite(rec_rel) and true_rel <= 100.0 * rec_rel + 1e-12
            return bool(ok), true_rel

    # ------------------------------------------------------------------ headline workload
    stage[0] = "matrix generation / upload"
    wl = build(a.workload, a.n, a.half_bandwidth, a.m, a.matrix, a.numbering)
    leg = Leg(wl)
    n, nnz_global, plan = wl["rows"], wl["nnz"], leg.plan
    note(f"matrix resident: {plan}")
    leg.ctx.spmv_bench(600)      # clocks and caches in steady state before the first timed iteration (the GPU idled while the
                                 # matrix was generated on the host: a 20-iteration region right after 2.5 ms of warm-up read 3 % high)
    stage[0] = "timed iterations"

    # main timed regions: exactly K iterations after W warm-up iterations each, no per-kernel instrumentation. `value` is the MEDIAN
    # region (an intermittent stall -- one region in a few dozen took 20 x as long in round 4, cause unknown -- must not become the
    # headline, nor be hidden: every region is in the line with the device's clock beside the host's).
    headline_regions = []

    def headline(lg):
        runs = []
        for _ in range(max(1, a.regions)):
            runs.append(lg.timed(a.method))
            headline_regions.append(lg.last_clocks)
        runs.sort(key=lambda r: r[0])
        return runs[(len(runs) - 1) // 2]

    dt, res = headline(leg)
    ok, true_relres = leg.check()
    if not ok and p2p_mode:
        # the peer-to-peer data path passed its self-test but not this check: measure with the collectives instead
        note(f"peer-to-peer data path failed the residual check (true relres {true_relres:.3e}); falling back")
        stage[0] = "fallback to the transport's collectives"
        failed_name = transport_name
        leg.close()
        barrier()
        L.bicg_comm_finalize()
        p2p_mode, transport_name = comm_setup(False)
        transport_name += f" (fallback: '{failed_name}' failed the residual check)"
        comm_info["fallback_reason"] = f"'{failed_name}' failed the residual check after the timed region (true relres {true_relres:.3e})"
        leg = Leg(wl)
        plan = leg.plan
        del headline_regions[:]
        dt, res = headline(leg)
        ok, true_relres = leg.check()
    head_flags_all = [k for k, v in leg.ctx.flags().items() if v]
    # multi-rank persistent launches: what the exchanges made the kernels wait (device clock, one sample per exchange)
    comm_info["wait_us"] = leg.ctx.comm_wait_stats() if world > 1 else None
    ms_step = 1e3 * dt / K
    relres = float(np.sqrt(res.dot_r / res.dot_zero)) if res.dot_zero > 0 else float("nan")
    genuine = res.iterations == W + K and np.isfinite(relres) and ok
    note(f"{a.method}: {ms_step:.4f} ms/iteration")
    if a.inner:       # the run rocprofv3 wraps: kernels only, no result line needed
        leg.close()
        dog.cancel()
        return

    stage[0] = "variant legs"
    head_flags = [k for k, v in leg.ctx.flags().items() if v]
    variants, variant_roof = {}, {}
    if not a.no_variants:
        for m in ITER_VECTOR_BYTES_PER_ROW:
            if m == a.method:
                variants[m] = ms_step
            else:
                dtv, _ = leg.best(m)
                variants[m] = 1e3 * dtv / K
            ib = iteration_bytes(m, nnz_global, n)
            gb = ib / (variants[m] * 1e-3) / 1e9
            variant_roof[m] = dict(ms_per_iteration=variants[m], algorithmic_bytes=ib, gbps=gb,
                                   **roof(ib, variants[m] * 1e-3, csr_bytes(nnz_global, n), leg.ctx.spmv_matrix_bytes(), 2, plan["rows"], NVEC[m], head_flags, stream, world))
    if not a.no_variants and a.workload == "transport":
        # BASELINE.json configs[4] family: 16 shifts, seed 7, sigma_j = (j+1) 0.01/16 (reference
        # src/main_shifted.c:99 pattern); 2 SpMV + one batched update over all shifts per iteration
        nsh, seed = 16, 7
        sigma = (np.arange(nsh) + 1.0) * 0.01 / nsh
        ks = min(K, 100)
        for which in ("shifted_lopbicgstab", "shifted_pipe_lopbicgstab"):
            rs = leg.ctx.solve_shifted(leg.b + sigma[seed] * leg.ones, sigma, seed, tol=0.0, max_iter=ks, check_every=ks, which=which)
            key = f"{which}_{nsh}shifts"
            variants[key] = 1e3 * rs["result"].seconds / max(rs["k"], 1)
            ib = shifted_iteration_bytes(nsh, nnz_global, n, "pipe" in which)
            gb = ib / (variants[key] * 1e-3) / 1e9
            variant_roof[key] = dict(ms_per_iteration=variants[key], algorithmic_bytes=ib, gbps=gb,
                                     **roof(ib, variants[key] * 1e-3, csr_bytes(nnz_global, n), leg.ctx.spmv_matrix_bytes(), 2, plan["rows"], 2 * nsh + 6, [], stream, world))
    # BASELINE.json configs[4] "batched SpMV": Y_j = (A + sigma_j I) X_j for 16 vectors in one pass over the matrix (the reference's
    # per-shift verification loop, src/test_shifted.c:129-154), device time of the pass
    spmm_leg = None
    if world == 1 and not a.no_variants and not a.inner and leg.ctx.flags().get("spmm"):
        stage[0] = "SpMM leg"
        nsh = 16
        X = np.random.default_rng(16).standard_normal((nsh, plan["rows"]))
        sg = (np.arange(nsh) + 1.0) * 0.01 / nsh
        leg.ctx.spmm(X, sg)
        ms_mm = min(leg.ctx.spmm(X, sg)[1] for _ in range(3))
        b8d = csr_bytes(nnz_global, n) + 2 * nsh * 8 * n
        bfmt = leg.ctx.spmv_matrix_bytes() + 2 * nsh * 8 * n
        spmm_leg = dict(ms=ms_mm, vectors=nsh, kernel=leg.ctx.last_spmm_kind(), bound="hbm", peak=HBM_PEAK_GBS,
                        frac=bfmt / (ms_mm * 1e-3) / 1e9 / HBM_PEAK_GBS, format_bytes=bfmt, algorithmic_bytes=b8d, survey_8d_bytes=b8d,
                        survey_8d_frac=b8d / (ms_mm * 1e-3) / 1e9 / HBM_PEAK_GBS, spmv_equivalents=None)
        del X
    spmv_alone_ms = leg.ctx.spmv_bench(200)
    if spmm_leg:
        spmm_leg["spmv_equivalents"] = spmm_leg["ms"] / spmv_alone_ms
        note(f"SpMM of 16 vectors: {1e3 * spmm_leg['ms']:.1f} us ({spmm_leg['kernel']}), {spmm_leg['spmv_equivalents']:.1f} x one product")
    # roofline leg: the same K iterations, every SpMV kernel launched with its own start/stop HIP events
    # (hipExtLaunchKernelGGL on the library's compute stream): kernel durations, no launch gaps. It comes after
    # the other timed legs of this matrix: event-timed launches put the queue into a profiling mode whose switch
    # back costs the next ordinary launches ~10 ms.
    stage[0] = "roofline leg"
    dt_ev, res_ev = leg.timed(a.method, kernel_events=True)
    spmv_ms = res_ev.spmv_ms_total / max(res_ev.spmv_launches, 1)
    b_spmv = spmv_bytes(plan["nnz_diag"] + plan["nnz_offd"], plan["rows"], plan["halo"])
    achieved = b_spmv / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    fmt_spmv = min(b_spmv, leg.ctx.spmv_matrix_bytes() + 16 * plan["rows"] + 16 * plan["halo"])
    fmt_gbps = fmt_spmv / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    leg.timed(a.method, steps=4, warm=0)       # back to ordinary launches before the next matrix is timed
    leg.close()
    comm_info["transport_used"] = transport_name

    # ------------------------------------------------------------------ how much of the headline is the synthetic's structure?
    # The Transport-SHAPED matrix has 15 constant diagonals: its slices are "uniform" (one shared list of distances, no column
    # index read: 8 instead of 10 bytes per non-zero). Transport.mtx itself (FEM) would not qualify. The same matrix once more
    # with that layout switched off (BICG_PLAN="uniform=0": 16-bit column offsets for every entry) puts a number on the difference.
    structure = None
    if world == 1 and not a.inner and not a.no_variants and a.workload == "transport" and not a.matrix:
        stage[0] = "headline without uniform slices"
        H.switches(uniform=0)
        try:
            lg = Leg(wl)
        finally:
            H.switches(uniform=None)
        dtu, _ = lg.best(a.method)
        _, rese = lg.timed(a.method, kernel_events=True)
        sp_u = rese.spmv_ms_total / max(rese.spmv_launches, 1)
        fb_u = lg.ctx.spmv_matrix_bytes() + 16 * lg.plan["rows"]
        structure = dict(setting="BICG_PLAN=uniform=0: every slice reads its 16-bit column offsets (10 bytes per non-zero)",
                         flags=[k for k, v in lg.ctx.flags().items() if v],
                         ms_per_iteration=1e3 * dtu / K, ms_per_iteration_default=ms_step,
                         spmv_avg_launch_ms=sp_u, spmv_avg_launch_ms_default=spmv_ms,
                         format_bytes_per_launch=fb_u, format_gbps=fb_u / (sp_u * 1e-3) / 1e9 if sp_u > 0 else None,
                         frac_of_format_bytes=fb_u / (sp_u * 1e-3) / 1e9 / HBM_PEAK_GBS if sp_u > 0 else None,
                         survey_8d_bytes_per_launch=b_spmv, survey_8d_frac=b_spmv / (sp_u * 1e-3) / 1e9 / HBM_PEAK_GBS if sp_u > 0 else None)
        lg.timed(a.method, steps=4, warm=0)
        lg.close()
        note(f"headline without uniform slices: {structure['ms_per_iteration']:.4f} ms/iteration (default {ms_step:.4f}), product {sp_u:.4f} ms (default {spmv_ms:.4f})")

    # ------------------------------------------------------------------ N > 1: the same matrix over RCCL collectives
    # north_star names "halo exchange and dot-product all-reduce on RCCL over xGMI": whichever path produced `value`,
    # the headline method is timed once more with the library's RCCL transport (grouped ncclSend/ncclRecv halo exchange,
    # packed ncclAllReduce per dot group, second stream above 6 M non-zeros per rank), so that a multi-GPU record says
    # what BOTH paths cost -- or why RCCL could not run.
    rccl_leg = None
    if world > 1 and not a.no_rccl_leg and not a.inner:
        stage[0] = "RCCL leg"
        if base_is_rccl(transport_name) and not p2p_mode:
            rccl_leg = {"ms_per_iteration": ms_step, "note": "the headline itself ran on the RCCL collectives"}
        elif world > torch.cuda.device_count() or a.transport in ("host", "host-p2p"):
            # ranks share a device (tests, the one-GPU box): RCCL is ASKED anyway, and what it answers goes into the record verbatim
            barrier()
            L.bicg_comm_finalize()
            _, rname = comm_setup(False, want="rccl")
            if comm_info["rccl_nranks"]:        # RCCL took two ranks on one device: time the leg like on a node
                lg = Leg(wl)
                dtr, resr = lg.best(a.method)
                okr, true_r = lg.check()
                rccl_leg = {"ms_per_iteration": 1e3 * dtr / K, "transport": rname, "rccl_nranks": comm_info["rccl_nranks"],
                            "iterations_genuine": bool(resr.iterations == W + K and okr), "true_relres_after_timed_region": true_r,
                            "note": "ranks SHARING one device"}
                lg.close()
            else:
                rccl_leg = {"unavailable": f"ranks share a device ({world} ranks, {torch.cuda.device_count()} GPU(s)): RCCL needs one GPU per rank",
                            "attempt": comm_info.get("rccl_error") or "refused on another rank"}
            barrier()
            L.bicg_comm_finalize()
            p2p_mode, _ = comm_setup(a.transport in ("auto", "host-p2p") and comm_info["fallback_reason"] is None, want=a.transport)
        else:
            barrier()
            L.bicg_comm_finalize()
            _, rname = comm_setup(False, want="rccl")
            if comm_info["rccl_nranks"]:
                lg = Leg(wl)
                dtr, resr = lg.best(a.method)
                okr, true_r = lg.check()
                rccl_leg = {"ms_per_iteration": 1e3 * dtr / K, "transport": rname, "rccl_nranks": comm_info["rccl_nranks"],
                            "iterations_genuine": bool(resr.iterations == W + K and okr), "true_relres_after_timed_region": true_r}
                lg.close()
            else:
                rccl_leg = {"unavailable": "the RCCL communicator could not be created on every rank (see stderr)"}
            # the extras below run on the transport that produced the headline
            barrier()
            L.bicg_comm_finalize()
            p2p_mode, _ = comm_setup(a.transport in ("auto", "host-p2p") and comm_info["fallback_reason"] is None)
        if rank == 0:
            note(f"RCCL leg: {rccl_leg}")

    # ------------------------------------------------------------------ the other workloads north_star names
    extras = {}
    if spmm_leg:
        extras["spmm_16_vectors"] = spmm_leg
    if not a.no_extras and a.workload == "transport" and not a.n and not a.matrix:
        def extra(name, wl2, methods, steps, kernel_roofline=False, spmm=False):
            stage[0] = f"extra workload {name}"
            lg = Leg(wl2)
            out = dict(rows=wl2["rows"], nnz=wl2["nnz"], workload=wl2["desc"], plan=lg.plan, flags=[k for k, v in lg.ctx.flags().items() if v],
                       setup_seconds=lg.setup_s)
            H.product_kernels()
            mbytes = lg.ctx.spmv_matrix_bytes()
            for m in methods:
                dtv, rv = lg.best(m, steps=steps, warm=min(W, 10))
                ms = 1e3 * dtv / steps
                ib = iteration_bytes(m, wl2["nnz"], wl2["rows"])
                # genuine: every timed iteration was an unconverged one (a converged or broken-down solve idles)
                out[m] = dict(ms_per_iteration=ms, algorithmic_bytes=ib, gbps=ib / (ms * 1e-3) / 1e9,
                              **roof(ib, ms * 1e-3, csr_bytes(wl2["nnz"], wl2["rows"]), mbytes, 2, lg.plan["rows"], NVEC[m], out["flags"], stream, world),
                              iterations=int(rv.iterations),
                              iterations_genuine=bool(int(rv.iterations) == steps + min(W, 10) and rv.breakdown_iteration == 0
                                                      and np.isfinite(rv.dot_r) and rv.dot_r > 0.0))
            out["product_kernels"] = H.product_kernels()       # which product kernels the solves of this leg launched (rank 0)
            sp = lg.ctx.spmv_bench(100)
            bs = spmv_bytes(lg.plan["nnz_diag"] + lg.plan["nnz_offd"], lg.plan["rows"], lg.plan["halo"])
            out["spmv_back_to_back"] = dict(ms=sp, gbps=bs / (sp * 1e-3) / 1e9, algorithmic_bytes_rank0=bs,
                                            **roof(bs, sp * 1e-3, csr_bytes(lg.plan["nnz_diag"] + lg.plan["nnz_offd"], lg.plan["rows"]), mbytes, 1, lg.plan["rows"], 2, out["flags"], stream, 1, spmv_only=True))
            if kernel_roofline:
                # the in-solver product of this matrix with its own start / stop events (as the headline's roofline leg)
                _, rese = lg.timed(methods[0], kernel_events=True, steps=steps, warm=min(W, 10))
                kms = rese.spmv_ms_total / max(rese.spmv_launches, 1)
                fb = min(bs, mbytes + 16 * lg.plan["rows"])
                out["kernel_roofline"] = dict(avg_launch_ms=kms, launches_timed=rese.spmv_launches,
                                              survey_8d_bytes_per_launch=bs, survey_8d_gbps=bs / (kms * 1e-3) / 1e9, survey_8d_frac=bs / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              format_bytes_per_launch=fb, format_gbps=fb / (kms * 1e-3) / 1e9, format_frac=fb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                lg.timed(methods[0], steps=4, warm=0)
            if spmm and lg.ctx.flags().get("spmm"):
                # BASELINE.json configs[4] "batched SpMV" on this matrix: 16 vectors in one pass over it (src/test_shifted.c:129-154)
                nsh = 16
                Xm = np.random.default_rng(16).standard_normal((nsh, lg.plan["rows"]))
                sgm = (np.arange(nsh) + 1.0) * 0.01 / nsh
                lg.ctx.spmm(Xm, sgm)
                msm = min(lg.ctx.spmm(Xm, sgm)[1] for _ in range(3))
                b8 = csr_bytes(wl2["nnz"], wl2["rows"]) + 2 * nsh * 8 * wl2["rows"]
                bf = mbytes + 2 * nsh * 8 * wl2["rows"]
                out["spmm_16_vectors"] = dict(ms=msm, vectors=nsh, kernel=lg.ctx.last_spmm_kind(), bound="hbm", peak=HBM_PEAK_GBS,
                                              frac=bf / (msm * 1e-3) / 1e9 / HBM_PEAK_GBS, format_bytes=bf, algorithmic_bytes=b8, survey_8d_bytes=b8,
                                              survey_8d_frac=b8 / (msm * 1e-3) / 1e9 / HBM_PEAK_GBS, spmv_equivalents=msm / sp)
                del Xm
            lg.close()
            note(f"{name}: " + ", ".join(f"{m} {out[m]['ms_per_iteration']:.4f} ms" for m in methods)
                 + (f", SpMM of 16 vectors {1e3 * out['spmm_16_vectors']['ms']:.0f} us ({out['spmm_16_vectors']['kernel']})" if "spmm_16_vectors" in out else ""))
            return out
        def laplace512():
            """BASELINE.json configs[3] at its stated size on ONE GPU (the N = 1 anchor of the 8-GPU configuration): 134 M rows,
            938 M non-zeros. The matrix is generated and planned on the device (bicg_stencil7_device, bicg_create_device_csr):
            no host copy exists; vectors (1 GB each) cross PCIe once."""
            stage[0] = "extra workload laplace7 512^3"
            m = 512
            rows, nnz = m ** 3, synth.stencil7_nnz(m)
            t0 = time.perf_counter()
            ctx, nnz_dev, plan_s, gen_s = H.Context.stencil7_on_device(m, synth.LAPLACE_WEIGHTS)
            assert nnz_dev == nnz
            out = dict(rows=rows, nnz=nnz, workload="BASELINE.json configs[3]: 7-point Laplacian 512^3 (134 M rows), generated and planned "
                       "on the GPU, b = A*1, x0 = 0, one MI355X", plan=ctx.plan_info(), flags=[k for k, v in ctx.flags().items() if v],
                       generate_seconds=gen_s, plan_seconds=plan_s, device_matrix_bytes=ctx.device_matrix_bytes(),
                       plane_marching_product=ctx.stencil_info())
            lg = Leg.__new__(Leg)
            lg.wl = dict(lo=0, hi=rows, desc="laplace7 512^3"); lg.ctx = ctx; lg.plan = out["plan"]
            lg.ones = np.ones(rows); lg.x0 = np.zeros(rows)
            lg.b = ctx.spmv(lg.ones)
            steps = min(K, 20)
            for mth in ("ca_bicgstab", "bicgstab"):
                dtv, rv = lg.best(mth, steps=steps, warm=3, tries=1)
                ms = 1e3 * dtv / steps
                ib = iteration_bytes(mth, nnz, rows)
                # the TRUE residual b - A x (one more product on the device) against the recursive one the iterations carried:
                # the check of the headline leg, at 134 M rows
                okv, true_rel = lg.check()
                out[mth] = dict(ms_per_iteration=ms, algorithmic_bytes=ib, gbps=ib / (ms * 1e-3) / 1e9,
                                **roof(ib, ms * 1e-3, csr_bytes(nnz, rows), ctx.spmv_matrix_bytes(), 2, rows, NVEC[mth], out["flags"], stream),
                                iterations=int(rv.iterations),
                                true_relres_after_timed_region=true_rel,
                                iterations_genuine=bool(int(rv.iterations) == steps + 3 and rv.breakdown_iteration == 0
                                                        and np.isfinite(rv.dot_r) and rv.dot_r > 0.0 and okv))
            sp = ctx.spmv_bench(20)
            bs = spmv_bytes(nnz, rows)
            out["spmv_back_to_back"] = dict(ms=sp, gbps=bs / (sp * 1e-3) / 1e9, algorithmic_bytes_rank0=bs,
                                            **roof(bs, sp * 1e-3, csr_bytes(nnz, rows), ctx.spmv_matrix_bytes(), 1, rows, 2, out["flags"], stream, 1, spmv_only=True))
            out["set_up_seconds_total"] = time.perf_counter() - t0
            ctx.close()
            note(f"laplace7 512^3: generated {gen_s:.2f} s, planned {plan_s:.2f} s, ca_bicgstab {out['ca_bicgstab']['ms_per_iteration']:.3f} ms, "
                 f"bicgstab {out['bicgstab']['ms_per_iteration']:.3f} ms per iteration")
            return out
        def small_rank_with_halo(nrows):
            """The 8-GPU form with traffic in it, on what one GPU can hold twice: TWO processes share this GPU, each holds half of a
            200 264-row Transport-shaped matrix (100 k rows + a 13 807-column halo towards the neighbour), pipelined BiCGStab as ONE
            persistent launch per chunk with the halo values pushed into the neighbour's landing ring and the dot sums crossing the
            mailboxes (this script under torch.distributed.run, transport host-p2p). Not a multi-GPU number: both ranks compete
            for the same CUs and fabric; what it measures is the LL protocol carrying real halos."""
            stage[0] = "extra workload small_rank_with_halo"
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", str(nrows), "--method", "pipe_bicgstab",
                   "--steps", "400", "--warmup", "40", "--transport", "host-p2p", "--no-cpu-baseline", "--no-variants", "--no-extras",
                   "--no-traffic", "--no-stream", "--no-rccl-leg", "--scale-decades", str(a.scale_decades)]
            child_full = tempfile.mktemp(prefix="bicg_bench_child_", suffix=".json", dir="/tmp")
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT,
                                   env=dict(os.environ, BENCH_WATCHDOG_S="240", OMP_NUM_THREADS="2", BENCH_FULL_JSON=child_full))
                assert r.stdout.startswith("{"), r.stderr[-300:]
                d = json.load(open(child_full))           # the child's complete record (its stdout line is the compact form)
                os.unlink(child_full)
                out = dict(ranks=2, rows_per_rank=nrows // 2, ms_per_iteration=d["value"], bound="latency", frac=None,
                           transport=d["config"]["transport"], flags=d["config"].get("flags"), halo=d["config"].get("halo"),
                           comm_wait_us=d.get("comm", {}).get("wait_us"),
                           iterations_genuine=d["config"]["iterations_genuine"],
                           true_relres_after_timed_region=d["config"]["true_relres_after_timed_region"],
                           note="two ranks SHARING this GPU (half the CUs each): the persistent kernel with real halos, not a 2-GPU time")
            except Exception as e:      # reported, never required
                out = {"error": repr(e)}
            note(f"small_rank_with_halo: {out}")
            return out
        ke = min(K, 100)
        for hb in (8, 64, 512):
            extras[f"banded_b{hb}"] = extra(f"banded b={hb}", build("banded", 0, hb), ("bicgstab", "pipe_bicgstab"), ke)
        if world > 1:
            # the unstructured matrix (RCM numbering) in the reference's row partition, whatever the number of GPUs
            extras["mesh_rcm"] = extra("mesh rcm", build("mesh", numbering="rcm"), ("bicgstab", "pipe_bicgstab"), ke)
        if world == 1:
            # what ONE of 8 GPUs does in BASELINE.json configs[2]: a 200 k-row rank (1/8 of the Transport-shaped matrix, the
            # reference's partition) -- latency-bound, the pipelined solver runs it as one persistent launch per chunk of
            # iterations (bicg_persist.hip). Single rank here: no halo, no cross-rank sums; tools/small_rank_times.sh drives
            # the same rank through the complete multi-rank path.
            n8 = (synth.TRANSPORT_N + 7) // 8
            wl8 = dict(build("transport", n=n8), desc=f"1/8 of the Transport-shaped matrix as one rank holds it at 8 GPUs ({n8} rows)")
            extras["transport_rank_of_8"] = extra("1/8 Transport rank", wl8, ("pipe_bicgstab", "bicgstab"), max(ke, 200))
            extras["small_rank_with_halo"] = small_rank_with_halo(n8)
            # ... and ONE of 4 / of 2 GPUs: 400 k rows (persistent launches with two rows per thread for all three methods since
            # round 6) and 800 k rows (the multi-launch iteration). Single rank, no link latency: the compute side of the strong-
            # scaling curve the driver measures with real GPUs.
            for parts in (4, 2):
                npart = (synth.TRANSPORT_N + parts - 1) // parts
                wlp = dict(build("transport", n=npart), desc=f"1/{parts} of the Transport-shaped matrix as one rank holds it at {parts} GPUs ({npart} rows)")
                extras[f"transport_rank_of_{parts}"] = extra(f"1/{parts} Transport rank", wlp, ("bicgstab", "pipe_bicgstab", "ca_bicgstab"), max(ke, 200))
            # the stand-in for Transport.mtx where nothing is regular: the unstructured FEM matrix in three numberings
            for kind in ("rcm", "generator", "random"):
                extras[f"mesh_{kind}"] = extra(f"mesh {kind}", build("mesh", numbering=kind), ("bicgstab", "pipe_bicgstab"), ke, kernel_roofline=True, spmm=kind != "random")
            extras["laplace7_256_ca"] = extra("laplace7 256^3", build("laplace7", m=256), ("ca_bicgstab", "bicgstab"), min(K, 50))
            extras["laplace7_256_ca"]["note"] = ("one GPU's share of BASELINE.json configs[3] (512^3 over 8 GPUs = 64 planes of 512^2 "
                                                 "= 16.8 M rows per GPU), CA-BiCGStab")
            if not a.no_laplace512:
                extras["laplace7_512_ca"] = laplace512()

    # ------------------------------------------------------------------ HBM traffic of this run's SpMV (rocprofv3 PMC)
    traffic, traffic_detail = None, None
    if rank == 0 and world == 1 and not a.no_traffic and a.workload == "transport" and not a.matrix:
        stage[0] = "rocprofv3 counter passes"
        inner = ["--inner", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-variants", "--no-extras", "--no-traffic",
                 "--method", a.method, "--scale-decades", str(a.scale_decades)] + (["--rows", str(a.n)] if a.n else [])
        m = measure_traffic(inner, note)
        if m:
            q_known = 2 * 8 * n
            factor = q_known / m["fplainq_fetch_counter_bytes"] if m.get("fplainq_fetch_counter_bytes") else 2.0
            traffic = 2.0 * m["fetch_counter_bytes"] + m["write_counter_bytes"]
            traffic_detail = dict(method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over a 30-iteration "
                                         "inner run of this command; KiB x 1024; FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64)",
                                  fetch_bytes_x2=2.0 * m["fetch_counter_bytes"], write_bytes=m["write_counter_bytes"], launches=m["launches"],
                                  fetch_factor_reproducing_k_vec_FPlainQ=factor,
                                  caveat="the x2 is stated by the guide for wide coalesced reads and re-checked here on the element-wise "
                                         "kernel; the SpMV issues 8- and 16-byte loads")

    # the product of the UNSTRUCTURED matrix as the second headline: the mesh matrix in RCM numbering (what a bandwidth-reducing
    # pre-pass of a FEM code hands over), with the generator order and the random permutation beside it
    KERNEL_TEXT = {"jagw": "k_spmv_jagw (csrc/bicg_jagw.hip: jagged slices, x window of each 256-row group in LDS, three dependent trips per group)",
                   "jagw_list": "k_spmv_jagl (csrc/bicg_jagw.hip: jagged slices, LIST-driven x window of each 256-row group in LDS, three trips)",
                   "jagd": "k_spmv_jagd (csrc/bicg_jagw.hip: jagged slices, x gathered through the caches by 16-bit offsets / 32-bit columns, three trips)",
                   "sell_jagged": "k_spmv_sell on jagged slices", "sell_window_loop": "k_spmv_sell's window loop", "csr": "k_spmv (CSR row blocks)"}

    def unstructured(kind, with_traffic):
        e = extras.get(f"mesh_{kind}")
        if not e or "kernel_roofline" not in e:
            return None
        kr = e["kernel_roofline"]
        t_un, t_un_detail = None, None
        if with_traffic and rank == 0 and world == 1 and not a.no_traffic:
            stage[0] = f"rocprofv3 counter passes, mesh matrix ({kind})"
            m = measure_traffic(["--inner", "--workload", "mesh", "--numbering", kind, "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-variants",
                                 "--no-extras", "--no-traffic", "--no-stream", "--method", "bicgstab", "--scale-decades", str(a.scale_decades)], note)
            if m:
                t_un = 2.0 * m["fetch_counter_bytes"] + m["write_counter_bytes"]
                t_un_detail = dict(fetch_bytes_x2=2.0 * m["fetch_counter_bytes"], write_bytes=m["write_counter_bytes"], launches=m["launches"])
        kern = [k for k in e.get("product_kernels", []) if k in KERNEL_TEXT]
        return dict(
            kernel="; ".join(KERNEL_TEXT[k] for k in kern) + f" on the unstructured FEM matrix, numbering '{kind}'",
            product_kernels=e.get("product_kernels"), numbering=kind,
            bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s",
            achieved=kr["format_gbps"], frac=kr["format_frac"], frac_basis="format bytes (values + 16-bit offsets or slots / 32-bit columns + per-lane words + x + y)",
            avg_launch_ms=kr["avg_launch_ms"], launches_timed=kr["launches_timed"],
            format_bytes_per_launch=kr["format_bytes_per_launch"], algorithmic_bytes_per_launch=kr["format_bytes_per_launch"],
            survey_8d_bytes_per_launch=kr["survey_8d_bytes_per_launch"], survey_8d_gbps=kr["survey_8d_gbps"], survey_8d_frac=kr["survey_8d_frac"],
            traffic=t_un, traffic_detail=t_un_detail,
            ms_per_iteration={m: e[m]["ms_per_iteration"] for m in ("bicgstab", "pipe_bicgstab") if m in e},
            back_to_back_spmv_ms=e["spmv_back_to_back"]["ms"],
            frac_of_measured_copy=(kr["format_gbps"] / stream["copy"]) if stream and "copy" in stream else None,
            survey_8d_frac_of_measured_copy=(kr["survey_8d_gbps"] / stream["copy"]) if stream and "copy" in stream else None)

    roofline_unstructured = unstructured("rcm", True)
    if roofline_unstructured:
        others = {}
        for kind in ("generator", "random"):
            u = unstructured(kind, kind == "random")
            if u:
                others[kind] = {k: u[k] for k in ("product_kernels", "avg_launch_ms", "frac", "survey_8d_frac", "survey_8d_frac_of_measured_copy", "traffic",
                                                  "format_bytes_per_launch", "ms_per_iteration")}
        # the adversarial numbering: every gather of x pulls a whole cache line out of the Infinity Cache for 8 useful bytes -- the
        # counters say how many bytes the product moved for the bytes its layout holds
        if "random" in others and others["random"]["traffic"]:
            others["random"]["bottleneck"] = ("x gathers: one cache line per entry from the Infinity Cache (x itself is 12.8 MB); fabric traffic = "
                                              f"{others['random']['traffic'] / others['random']['format_bytes_per_launch']:.1f} x the layout's bytes")
        roofline_unstructured["other_numberings"] = others

    cpu = None
    cpu_all = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload == "transport" and not a.matrix:
        stage[0] = "cpu baseline"
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--n", str(n),
                                  "--scale-decades", str(a.scale_decades), "--iters", str(a.cpu_iters),
                                  "--method", a.method], capture_output=True, text=True, timeout=600)
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
        iter_bytes = iteration_bytes(a.method, nnz_global, n)
        iter_fmt_bytes = min(iter_bytes, iter_bytes - 2 * world * max(0, b_spmv - fmt_spmv))
        shape = {"transport": "Transport-shaped CSR", "laplace7": f"7-point Laplacian {a.m}^3", "banded": f"banded CSR b={a.half_bandwidth}",
                 "fem_like": "FEM-like irregular CSR", "mesh": f"unstructured FEM CSR ({a.numbering} numbering)"}[a.workload]
        line = {
            "metric": f"ms/iteration, {a.method}, {shape} ({n} rows, {nnz_global} nnz), strong scaling over GPUs",
            "value": ms_step, "unit": "ms/iteration", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_step, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": wl["data"],
            "config": {"workload": wl["desc"],
                       "rows": n, "nnz": nnz_global, "scale_decades": a.scale_decades, "method": a.method,
                       "partition": f"row blocks over {world} GPU(s), reference src/matrix.c:295-308",
                       "transport": transport_name, "flags": head_flags_all, "halo": int(plan["halo"]),
                       "iterations_genuine": bool(genuine), "relres_after_timed_region": relres,
                       "true_relres_after_timed_region": true_relres,
                       "setup_seconds_rank0": leg.setup_s},
            # bytes one iteration has to move with the stored layout (2 products + the fused-minimum vector traffic of SURVEY.md 8d)
            # (keys as in rounds 1-3: the SURVEY.md 8d figure -- 2 CSR products + the fused-minimum vector traffic; the bytes of the
            # stored layout have keys of their own)
            "hbm_gbps_iteration": iter_bytes / (ms_step * 1e-3) / 1e9,
            "iteration_algorithmic_bytes": iter_bytes,
            "iteration_format_bytes": iter_fmt_bytes,
            "hbm_gbps_iteration_format": iter_fmt_bytes / (ms_step * 1e-3) / 1e9,
            "headline_regions": headline_regions,
            "headline_regions_note": f"{len(headline_regions)} regions of exactly {K} iterations; value = the median region; per region the host's "
                                     "bracket (barrier + synchronize on both sides), the device's events around the launches, the host time "
                                     "spent enqueueing, the library's own wall time",
            # `achieved` / `frac`: the ALGORITHMIC bytes of one launch of this kernel on this matrix -- what the stored layout has to
            # move (values + the index the layout keeps, 16-bit distances for slices that are not uniform, none for
            # those that are, + row lengths + x + y; `traffic` from the counters agrees within 2 %) -- over the launch time. The CSR
            # figure of SURVEY.md 8d (12 bytes per non-zero) stays beside it: a layout that stores no column index moves fewer bytes
            # than CSR, and the CSR bytes over the launch time (`csr_equivalent_gbps`) can exceed the HBM peak.
            "roofline": {"kernel": "k_spmv_sell (sliced-ELL SpMV with fused dot epilogue), rank 0 share", "bound": "hbm",
                         "achieved": fmt_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fmt_gbps / HBM_PEAK_GBS,
                         "frac_basis": "format bytes: what the stored layout has to move (uniform slices read no column index); the SURVEY.md "
                                       "8d figure (12 bytes per non-zero) is survey_8d_frac, the same matrix without uniform slices is "
                                       "structure_dependence",
                         "survey_8d_bytes_per_launch": b_spmv, "survey_8d_gbps": achieved, "survey_8d_frac": achieved / HBM_PEAK_GBS,
                         "structure_dependence": structure,
                         "traffic": traffic, "traffic_detail": traffic_detail, "algorithmic_bytes_per_launch": fmt_spmv,
                         "avg_launch_ms": spmv_ms, "launches_timed": res_ev.spmv_launches,
                         "csr_bytes_per_launch": b_spmv, "csr_equivalent_gbps": achieved, "csr_equivalent_frac": achieved / HBM_PEAK_GBS,
                         "format_bytes_per_launch": fmt_spmv, "format_gbps": fmt_gbps, "frac_of_format_bytes": fmt_gbps / HBM_PEAK_GBS,
                         "ms_per_step_with_events": 1e3 * dt_ev / K,
                         "back_to_back_spmv_ms": spmv_alone_ms,
                         # the denominators north_star asks for, measured on THIS GPU in this run (stream leg above)
                         "stream_measured_gbps": stream,
                         # (of the bytes the layout moves: the CSR figure over a measured rate would exceed 1)
                         "frac_of_measured_stream": (fmt_gbps / stream["triad"]) if stream and "triad" in stream else None,
                         "frac_of_measured_copy": (fmt_gbps / stream["copy"]) if stream and "copy" in stream else None,
                         "frac_of_measured_read": (fmt_gbps / stream["read8"]) if stream and "read8" in stream else None},
            "roofline_unstructured": roofline_unstructured,
            "comm": dict(comm_info, rccl_leg=rccl_leg),
            "cpu_baseline": cpu,
            "cpu_baseline_multicore": cpu_all,
            "variants_ms_per_iteration": variants,
            "variant_rooflines": variant_roof,
            "extras": extras,
            "timed_regions_ms": regions_ms,
            "timed_region_clocks": region_clocks,
        }
        # the complete record (every timed region with its clocks, plans, flags, notes): a side file and stderr. stdout gets
        # ONE compact line (< 8 KB) formed from it -- the driver's capture could not hold the 27 KB line of round 5
        full_path = os.environ.get("BENCH_FULL_JSON", os.path.join(ROOT, "bench_full.json"))
        line["full_record"] = os.path.basename(full_path)
        try:
            with open(full_path, "w") as f:
                json.dump(line, f, indent=1)
            out_dir = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(out_dir) and "BENCH_FULL_JSON" not in os.environ:
                shutil.copy(full_path, os.path.join(out_dir, "bench_full.json"))
        except OSError as e:
            note(f"bench_full.json not written: {e!r}")
        # (not echoed to stderr: the driver keeps a bounded tail of stdout + stderr TOGETHER, a 27 KB echo would push the line out)
        print(compact(line), file=result_out, flush=True)

    dog.cancel()
    if dist is not None:
        dist.barrier()
        L.bicg_comm_finalize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
