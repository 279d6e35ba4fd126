#!/bin/bash
# round 5, third GPU call: tiled element-wise kernels at 512^3 (A/B + kernel stats), counters of the FEM-like product, the new
# drop-in diagnostics test, 200 consecutive regions with three clocks
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c3
rm -rf $OUT; mkdir -p $OUT
cd $R
for env in "BICG_VEC_TILE=1" "BICG_VEC_TILE=0"; do
  env $env timeout 200 python tools/stencil_sweep.py 512 "" > $OUT/sweep512_$env.txt 2>&1; cat $OUT/sweep512_$env.txt
done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats512 -o s --output-format csv -- python $R/tools/lap512_only.py > $OUT/stats512.log 2>&1
cp $(find $OUT/stats512 -name "s_kernel_stats.csv" | head -1) $OUT/laplace512_kernel_stats.csv 2>/dev/null
head -20 $OUT/laplace512_kernel_stats.csv; tail -4 $OUT/stats512.log
cd $R
SPMV_KIND=fem_like bash tools/r5/pmc_spmv.sh r5c3/pmc_fem SPMV_KIND=fem_like > $OUT/pmc_fem_stdout.txt 2>&1; tail -60 $OUT/pmc_fem_stdout.txt
timeout 600 python -m pytest tests/test_dropin_host.py -x -q -k "display_error" 2>&1 | tail -15 > $OUT/pytest_display_error.txt; cat $OUT/pytest_display_error.txt
timeout 300 python tools/region_log.py 200 20 5 > $OUT/regions_200.txt 2>&1; tail -8 $OUT/regions_200.txt
