#!/usr/bin/env python3
"""bench.py's variant legs in bench.py's order and with its calls (5 warm-up iterations, then ONE call of 20, check_every 20),
every call timed on the host; then the same solver again iteration by iteration. The first timed region of a solver that has not
run in the process before is the one that stalled in rounds 2-4 (bench.py `timed_regions_ms`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
if os.environ.get("STALL_TORCH") == "1":          # what bench.py has in its process: torch's HIP runtime, a timer thread, a second stdout
    import threading, torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
    dog = threading.Timer(900, lambda: None); dog.daemon = True; dog.start()
    junk = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(scale_decades=2.0)
ctx = H.Context(H.single_rank_blocks(A))
b = ctx.spmv(np.ones(A.rows))
x0 = np.zeros(A.rows)
single = os.environ.get("STALL_SINGLE_STEPS") == "1"
for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr"):
    for rep in range(2):
        ctx.load(x0, b)
        ctx.run_begin(method, tol=0.0, max_iter=25, check_every=20, krr=50, nrr=2)
        t = time.perf_counter(); ctx.run_iterate(5); ctx.sync(); warm = 1e3 * (time.perf_counter() - t)
        if os.environ.get("STALL_TORCH") == "1":
            torch.cuda.synchronize()
        if single:
            ms = []
            for _ in range(20):
                t = time.perf_counter(); ctx.run_iterate(1); ctx.sync(); ms.append(1e3 * (time.perf_counter() - t))
            region = " ".join("%.2f" % v for v in ms)
        else:
            t = time.perf_counter(); ctx.run_iterate(20); ctx.sync(); region = "%.2f" % (1e3 * (time.perf_counter() - t))
        t = time.perf_counter(); ctx.run_end(); end = 1e3 * (time.perf_counter() - t)
        print("%-18s try %d: warm-up (5) %.2f ms, timed region (20): %s ms, run_end %.2f ms" % (method, rep, warm, region, end), flush=True)
ctx.close()
