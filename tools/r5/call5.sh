#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c5
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_full_size.py -x -q -k "ragged_rows_product or jagged_slices" 2>&1 | tail -5
python - > $OUT/fem_times.txt 2>&1 <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
A = synth.fem_like(scale_decades=2.0)
for env in ({"BICG_JAGW": "0"}, {"BICG_JAGW": "1"}):
    os.environ.update(env)
    ctx = H.Context(H.single_rank_blocks(A))
    b = ctx.spmv(np.ones(A.rows))
    sp = min(ctx.spmv_bench(100) for _ in range(3))
    out = {}
    for method in ("bicgstab", "pipe_bicgstab"):
        best = 1e9
        for rep in range(3):
            ctx.load(np.zeros(A.rows), b)
            ctx.run_begin(method, tol=0.0, max_iter=110, check_every=100)
            ctx.run_iterate(10); ctx.sync()
            t = time.perf_counter(); ctx.run_iterate(100); ctx.sync(); best = min(best, (time.perf_counter() - t) / 100 * 1e3)
            ctx.run_end()
        out[method] = best
    print(env, "product back to back %.2f us, plain %.4f, pipelined %.4f ms/iteration" % (sp * 1e3, out["bicgstab"], out["pipe_bicgstab"]), flush=True)
    ctx.close()
PY
cat $OUT/fem_times.txt
