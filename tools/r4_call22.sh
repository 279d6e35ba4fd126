#!/bin/bash
# round 4, GPU call 22: where the set-up goes (BICG_PLAN_TRACE) -- Transport-shaped and FEM-like blocks through bicg_create
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c22
mkdir -p $OUT
cd $R
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
BICG_PLAN_TRACE=1 timeout 300 python - > $OUT/plan_trace.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
for name, A in (("transport_like", synth.transport_like(scale_decades=2.0)), ("fem_like", synth.fem_like(scale_decades=2.0))):
    blocks = H.single_rank_blocks(A)
    for rep in range(2):
        t = time.perf_counter()
        ctx = H.Context(blocks)
        print(name, "bicg_create %.4f s" % (time.perf_counter() - t), flush=True)
        ctx.close()
PY
cat $OUT/host.txt; cat $OUT/plan_trace.txt
