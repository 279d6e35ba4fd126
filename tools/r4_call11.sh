#!/bin/bash
# round 4, GPU call 11: contiguous multi-group workgroups + plane-block order of the groups on the 512^3 Laplacian (A/B), windowed
# SpMM with one staging round, the tests around the changed product kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c11
mkdir -p $OUT
cd $R
timeout 200 python tools/spmm_only.py > $OUT/spmm_only.txt 2>&1
timeout 300 python tools/lap512_only.py > $OUT/lap512.txt 2>&1
BICG_SELL_BLOCK=0 timeout 300 python tools/lap512_only.py > $OUT/lap512_natural_order.txt 2>&1
BICG_SELL_BLOCK=128 timeout 300 python tools/lap512_only.py > $OUT/lap512_block128.txt 2>&1
BICG_SELL_BLOCK=0 BICG_SELL_GPW=1 BICG_SELL_GPW_DOTS=1 timeout 300 python tools/lap512_only.py > $OUT/lap512_one_group_per_wg.txt 2>&1
timeout 1200 python -m pytest -q --durations=8 -m gpu tests/test_gpu_parity.py tests/test_full_size.py tests/test_bench_workloads.py \
   "tests/test_multirank_fullsize.py" tests/test_shifted.py > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab timeout 300 python tools/ab.py "" > $OUT/ab_default.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt | tail -20
for f in spmm_only lap512 lap512_natural_order lap512_block128 lap512_one_group_per_wg ab_default; do echo "== $f"; cat $OUT/$f.txt; done
