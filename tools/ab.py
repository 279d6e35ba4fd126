#!/usr/bin/env python3
"""In-process A/B of library configurations (environment read at bicg_create): each argument is a
';'-separated list of NAME=VALUE; contexts are created once and timed alternately.
   python tools/ab.py "" "BICG_SPMV_VARIANT=16" "BICG_SELL_GPW_DOTS=1" """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
if os.environ.get("AB_MATRIX") == "fem_like":          # the irregular matrix of bench.py's extras
    A = synth.fem_like(scale_decades=2.0)
else:
    A = synth.transport_like(n=int(os.environ.get("AB_N", synth.TRANSPORT_N)), scale_decades=2.0)
methods = os.environ.get("AB_METHODS", "bicgstab").split(",")
ctxs = []
for spec in sys.argv[1:]:
    saved = dict(os.environ)
    for kv in filter(None, spec.split(";")):
        k, v = kv.split("="); os.environ[k] = v
    ctxs.append((spec or "default", H.Context(H.single_rank_blocks(A))))
    os.environ.clear(); os.environ.update(saved)
b = ctxs[0][1].spmv(np.ones(A.rows))
best = {}
for rep in range(int(os.environ.get("AB_REPS", "4"))):
    for name, ctx in ctxs:
        for m in methods:
            ctx.load(np.zeros(A.rows), b)
            ctx.run_begin(m, tol=0.0, max_iter=220, check_every=200)
            ctx.run_iterate(20)
            t = time.perf_counter(); ctx.run_iterate(200); dt = (time.perf_counter() - t) / 200 * 1e6
            ctx.run_end()
            best[(name, m)] = min(best.get((name, m), 1e9), dt)
        best[(name, "spmv")] = min(best.get((name, "spmv"), 1e9), ctx.spmv_bench(50) * 1e3)
for name, _ in ctxs:
    print(f"{name:45s} " + "  ".join(f"{m} {best[(name, m)]:.1f}us" for m in methods + ["spmv"]), flush=True)
