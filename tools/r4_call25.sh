#!/bin/bash
# round 4, GPU call 25: the four wavefronts of a workgroup on the same x segment of four consecutive grid lines (SellDev::ystride)
# against four consecutive slices: 512^3 / 256^3 products and solvers, parity of the Laplacian tests
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c25
mkdir -p $OUT
cd $R
BICG_SELL_YGROUP=0 timeout 100 python tools/lap512_spmv.py > $OUT/spmv512_consecutive.txt 2>&1
timeout 100 python tools/lap512_spmv.py > $OUT/spmv512_ygroup.txt 2>&1
BICG_SELL_BLOCK=0 timeout 100 python tools/lap512_spmv.py > $OUT/spmv512_ygroup_natural.txt 2>&1
BICG_SELL_BLOCK=1 timeout 100 python tools/lap512_spmv.py > $OUT/spmv512_ygroup_b1.txt 2>&1
BICG_SELL_BLOCK=32 timeout 100 python tools/lap512_spmv.py > $OUT/spmv512_ygroup_b32.txt 2>&1
tail -n 1 $OUT/spmv512_*.txt
timeout 200 python tools/lap512_only.py > $OUT/lap512_ygroup.txt 2>&1
timeout 100 python tools/lap512_only.py 256 > $OUT/lap256_ygroup.txt 2>&1
tail -n 3 $OUT/lap512_ygroup.txt $OUT/lap256_ygroup.txt
timeout 900 python -m pytest -q -m gpu tests/test_full_size.py tests/test_bench_workloads.py -k "laplace or device_side or list_driven" > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt | tail
