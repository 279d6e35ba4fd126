#!/usr/bin/env python3
"""The m^3 7-point Laplacian (BASELINE.json configs[3] for m = 512) generated and planned on the device, once per setting of the
plane-marching product's knobs: product back to back, plain and CA-BiCGStab per iteration.
    python tools/stencil_sweep.py m "lines=2 planes=8" "stencil=0" "BICG_SELL_XCD=0" ...      ("" = defaults; lower case:
tokens of BICG_PLAN, upper case: variables -- the measurement knobs need a library built with make EXPERIMENTS=1)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
m = int(sys.argv[1])
n = m ** 3
KNOBS = ("BICG_PLAN", "BICG_SELL_XCD", "BICG_SELL_ALT", "BICG_STENCIL_XCD", "BICG_STENCIL_NT", "BICG_STENCIL_LDS")
for setting in sys.argv[2:] or [""]:
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in setting.split():
        k, v = kv.split("=")
        if k in H.SWITCHES:
            H.switches(**{k: v})
        else:
            os.environ[k] = v
    ctx, nnz, ps, gs = H.Context.stencil7_on_device(m, synth.LAPLACE_WEIGHTS)
    info = ctx.stencil_info()
    b = ctx.spmv(np.ones(n))
    out = {}
    for method in () if os.environ.get("PRODUCT_ONLY") else ("bicgstab", "ca_bicgstab"):
        best = 1e9
        for rep in range(2):
            ctx.load(np.zeros(n), b)
            ctx.run_begin(method, tol=0.0, max_iter=25, check_every=25)
            ctx.run_iterate(5); ctx.sync()
            t = time.perf_counter(); ctx.run_iterate(20); ctx.sync(); dt = (time.perf_counter() - t) / 20
            ctx.run_end()
            best = min(best, dt * 1e3)
        out[method] = best
    sp = ctx.spmv_bench(20)
    print("%d^3 [%-40s] product %.4f ms  plain %.4f  CA %.4f ms/iteration | plan %.2f s, stencil %s, matrix-side bytes per product %d"
          % (m, setting, sp, out.get("bicgstab", 0.0), out.get("ca_bicgstab", 0.0), ps, info, ctx.spmv_matrix_bytes()), flush=True)
    ctx.close()
