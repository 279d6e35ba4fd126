#!/usr/bin/env python3
"""Where k_spmm_jpipe's time goes (csrc/bicg_spmm_jag.hip): BICG_TEST=spmm-skip=n switches parts of the kernel off (the results are
then wrong): 1 no staging loads, 2 no head products, 16 no tail products, 8 no LDS stores of the window; further tokens
(spmm-jbuf, spmm-jres) select the shape. One matrix, one context, 16 vectors, the kernel's own start / end events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
kind = os.environ.get("SPMM_MATRIX", "mesh_rcm")
if kind.startswith("mesh_"):
    from mpi_bicgstab_amd import mesh
    import tempfile
    cache = os.path.join(tempfile.gettempdir(), "bicg_mesh_cache")      # (bench.py's: generating the mesh under rocprofv3 takes minutes)
    os.makedirs(cache, exist_ok=True)
    A = mesh.fem_unstructured(117, kind[5:], 2.0, cache_dir=cache)
elif kind == "fem_like":
    A = synth.fem_like(scale_decades=2.0)
else:
    A = synth.transport_like(scale_decades=2.0)
X = np.random.default_rng(0).standard_normal((16, A.rows))
sg = (np.arange(16) + 1.0) * 0.01 / 16
ctx = H.Context(H.single_rank_blocks(A))
toks = os.environ.get("SPMM_TOKENS", "spmm-skip=0;spmm-skip=1;spmm-skip=2;spmm-skip=16;spmm-skip=18;spmm-skip=8;spmm-skip=9;spmm-skip=27").split(";")
for t in toks:
    os.environ["BICG_TEST"] = t
    best = 1e9
    for _ in range(6):
        _, ms = ctx.spmm(X, sg)
        best = min(best, ms)
    print(kind, t, "kind", int(H.lib().bicg_last_spmm_windowed(ctx.h)), "best of 6: %.1f us" % (1e3 * best), flush=True)
ctx.close()
