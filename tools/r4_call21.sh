#!/bin/bash
# round 4, GPU call 21: the loop of its own for blocks whose slices are all list-driven (SellDev::all_lists): 512^3 / 256^3 with and
# without it, parity of the Laplacian tests (single GPU, 8 ranks), the full-size and bench-workload tests
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c21
mkdir -p $OUT
cd $R
BICG_SELL_LISTS=0 timeout 200 python tools/lap512_only.py > $OUT/lap512_nolists.txt 2>&1
timeout 200 python tools/lap512_only.py > $OUT/lap512_lists.txt 2>&1
timeout 100 python tools/lap512_only.py 256 > $OUT/lap256_lists.txt 2>&1
timeout 900 python -m pytest -q -m gpu tests/test_full_size.py tests/test_bench_workloads.py tests/test_multirank_fullsize.py -k "laplace or device_side" > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt | tail; tail -n 3 $OUT/lap512_nolists.txt $OUT/lap512_lists.txt $OUT/lap256_lists.txt
