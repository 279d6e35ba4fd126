#!/bin/bash
# round 4, GPU call 14: masked slices (the slices next to a grid face: one 16-bit word per row) -- 512^3 with / without, parity
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c14
mkdir -p $OUT
cd $R
timeout 300 python tools/lap512_only.py > $OUT/lap512.txt 2>&1
BICG_SELL_MASKED=0 timeout 300 python tools/lap512_only.py > $OUT/lap512_no_masked.txt 2>&1
timeout 300 python tools/lap512_only.py 256 > $OUT/lap256.txt 2>&1
timeout 900 python -m pytest -q -m gpu tests/test_gpu_parity.py tests/test_full_size.py "tests/test_bench_workloads.py::test_laplace7_256_as_benchmarked" > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt | tail; cat $OUT/lap512.txt $OUT/lap512_no_masked.txt $OUT/lap256.txt
