#!/bin/bash
# round 4, GPU call 3: new defaults (alternating + XCD-contiguous products, slots by row group); all-NT vector streams; tiles
# instead of grid-stride in the element-wise kernels; counters of the in-solver product
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4c3
mkdir -p $OUT
timeout 600 python -m pytest -x -q tests/test_gpu_parity.py tests/test_full_size.py tests/test_shifted.py tests/test_switching.py "tests/test_dropin_host.py" > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
for spec in "" "BICG_VEC_NT=1" "BICG_VEC_NT=3" "BICG_VEC_NT=2" "BICG_VEC_PPT=1" "BICG_VEC_PPT=2" "BICG_VEC_PPT=4" "BICG_VEC_PPT=8" "BICG_VEC_GRID=1024" "BICG_VEC_NT=1;BICG_VEC_PPT=4" "BICG_SELL_ALT=0;BICG_SELL_XCD=0"; do
  AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab timeout 200 python tools/ab.py "$spec" >> $OUT/ab_vec.txt 2>&1
done
# 512^3 and 256^3: element-wise kernels as tiles
for spec in "" "BICG_VEC_PPT=4" "BICG_VEC_PPT=8"; do
  echo "== $spec" >> $OUT/lap.txt
  env $(echo $spec | tr ';' ' ') timeout 300 python - >> $OUT/lap.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
for m in (256, 512):
    ctx, nnz, ps, gs = H.Context.stencil7_on_device(m, synth.LAPLACE_WEIGHTS)
    n = m ** 3
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
done
bash tools/r4_pmc.sh > $OUT/pmc_log.txt 2>&1
cp gpurun_out/r4pmc/summary.txt $OUT/pmc_summary.txt
tail -5 $OUT/tests.txt; cat $OUT/ab_vec.txt $OUT/lap.txt; cat $OUT/pmc_summary.txt
