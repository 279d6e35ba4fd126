#!/usr/bin/env python3
"""The FEM-like matrix of bench.py (ragged rows 6..27, 1.6 M rows: what an unstructured matrix such as Transport.mtx gets): product
back to back, plain / CA / pipelined BiCGStab per iteration, once per setting.
    python tools/fem_like_times.py "" "jagw=0" ...        (lower case: tokens of BICG_PLAN, upper case: variables;
    BICG_HIP_LIB=<path> selects another build of the library for an A/B on one box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.fem_like(scale_decades=2.0)       # as bench.py generates it: unconverged over the timed iterations
n = A.rows
tag = os.path.basename(os.environ.get("BICG_HIP_LIB", "libbicgstab_hip.so"))
for setting in sys.argv[1:] or [""]:
    saved = {k: os.environ.get(k) for k in H.SWITCH_VARS}
    extra = []
    for kv in setting.split():
        k, v = kv.split("=")
        if k in H.SWITCHES:
            H.switches(**{k: v})
        else:
            os.environ[k] = v; extra.append(k)
    ctx = H.Context(H.single_rank_blocks(A))
    b = ctx.spmv(np.ones(n))
    out = {}
    for method in ("bicgstab", "ca_bicgstab", "pipe_bicgstab"):
        best = 1e9
        for rep in range(3):
            ctx.load(np.zeros(n), b)
            ctx.run_begin(method, tol=0.0, max_iter=130, check_every=130)
            ctx.run_iterate(20); ctx.sync()
            t = time.perf_counter(); ctx.run_iterate(100); ctx.sync(); dt = (time.perf_counter() - t) / 100
            ctx.run_end()
            best = min(best, dt * 1e3)
        out[method] = best
    sp = min(ctx.spmv_bench(100) for _ in range(3))
    print("%s [%-24s] product %.2f us  plain %.4f  CA %.4f  pipelined %.4f ms/iteration" % (tag, setting, sp * 1e3, out["bicgstab"], out["ca_bicgstab"], out["pipe_bicgstab"]), flush=True)
    ctx.close()
    for k in extra:
        os.environ.pop(k, None)
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
