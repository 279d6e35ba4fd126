#!/bin/bash
# round 4, GPU call 4: uniform slices (no column traffic in the interior of banded / stencil matrices)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4c4
mkdir -p $OUT
timeout 900 python -m pytest -x -q --durations=8 tests/test_gpu_parity.py tests/test_full_size.py tests/test_bench_workloads.py tests/test_dropin_host.py > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab timeout 300 python tools/ab.py "" "BICG_SELL_UNIFORM=0" > $OUT/ab_uniform.txt 2>&1
timeout 300 python - > $OUT/lap.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
for m in (256, 512):
    ctx, nnz, ps, gs = H.Context.stencil7_on_device(m, synth.LAPLACE_WEIGHTS)
    n = m ** 3
    print(m, "plan", ps, "s  uniform entries", ctx.uniform_entries(), "of", nnz, "spmv matrix bytes", ctx.spmv_matrix_bytes(), flush=True)
    b = ctx.spmv(np.ones(n))
    for method in ("bicgstab", "ca_bicgstab"):
        ctx.load(np.zeros(n), b)
        ctx.run_begin(method, tol=0.0, max_iter=25, check_every=25)
        ctx.run_iterate(5); ctx.sync()
        t = time.perf_counter(); ctx.run_iterate(20); ctx.sync(); dt = (time.perf_counter() - t) / 20
        ctx.run_end()
        print(m, method, "%.3f ms per iteration" % (dt * 1e3), flush=True)
    print(m, "spmv back to back %.3f ms" % ctx.spmv_bench(20), flush=True)
    ctx.close()
PY
for spec in "400528 pipe_bicgstab" "801056 pipe_bicgstab"; do
  echo "=== $spec" >> $OUT/persist_check.txt
  timeout 200 python tools/persist_check.py $spec >> $OUT/persist_check.txt 2>&1
done
tail -12 $OUT/tests.txt; cat $OUT/ab_uniform.txt $OUT/lap.txt; grep "us per iteration\|===" $OUT/persist_check.txt
