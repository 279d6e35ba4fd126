#!/usr/bin/env python3
"""N consecutive timed regions of bench.py's headline leg (Transport-shaped matrix, W warm-up + K timed iterations each, the four
solvers in turn), every region with three clocks: the host's bracket around the K iterations, the device's events around the
launches, the host time spent enqueueing (bicg_run_iterate_timed). A region that is long on the host's clock only was held up
outside the kernels. Prints one line per region and a summary per solver (median, maximum, regions beyond 2 x the median).
    python tools/region_log.py [regions=200] [K=20] [W=5]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_bicgstab_amd import hipsolver as H, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
b = ctx.spmv(np.ones(A.rows)); x0 = np.zeros(A.rows)
ctx.spmv_bench(600)
methods = ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr")
log = {m: [] for m in methods}
print("region method host_ms device_ms enqueue_ms library_wall_ms")
for i in range(N):
    m = methods[i % len(methods)]
    ctx.load(x0, b)
    ctx.run_begin(m, tol=0.0, max_iter=W + K, check_every=max(W, K, 1), krr=50, nrr=2)
    ctx.run_iterate(W); ctx.sync()
    t0 = time.perf_counter()
    _, c = ctx.run_iterate_timed(K)
    ctx.sync()
    host = 1e3 * (time.perf_counter() - t0)
    ctx.run_end()
    log[m].append((host, c["device_ms"], c["enqueue_ms"], c["wall_ms"]))
    print(f"{i:4d} {m:18s} {host:9.4f} {c['device_ms']:9.4f} {c['enqueue_ms']:9.4f} {c['wall_ms']:9.4f}", flush=True)
print()
for m in methods:
    a = np.array(log[m])
    med = np.median(a[:, 0])
    out = [(i, *row) for i, row in enumerate(a) if row[0] > 2.0 * med]
    print(f"{m:18s} regions {len(a):3d}  host median {med:.4f} ms  max {a[:, 0].max():.4f}  device median {np.median(a[:, 1]):.4f}  max {a[:, 1].max():.4f}  "
          f"enqueue median {np.median(a[:, 2]):.4f}  regions beyond 2 x median: {len(out)} {out}")
ctx.close()
