#!/bin/bash
# round 4, GPU call 19: (a) list-driven slices by descriptor (SellDev::sdesc) with the next slices' metadata requested ahead:
# 512^3 / 256^3 Laplacian with and without; (b) direct SpMM (k_spmm_dir) against the windowed one; parity of both
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c19
mkdir -p $OUT
cd $R
BICG_SELL_DESC=0 timeout 200 python tools/lap512_only.py > $OUT/lap512_nodesc.txt 2>&1
timeout 200 python tools/lap512_only.py > $OUT/lap512_desc.txt 2>&1
timeout 100 python tools/lap512_only.py 256 > $OUT/lap256_desc.txt 2>&1
BICG_SPMM_WIN=1 timeout 100 python tools/spmm_only.py > $OUT/spmm_win.txt 2>&1
BICG_SPMM_WIN=2 timeout 100 python tools/spmm_only.py > $OUT/spmm_dir.txt 2>&1
timeout 600 python -m pytest -q -m gpu tests/test_full_size.py tests/test_bench_workloads.py -k "laplace or device_side or spmm or config5" > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
BICG_SPMM_WIN=2 timeout 300 python -m pytest -q -m gpu tests/test_full_size.py tests/test_shifted.py tests/test_bench_workloads.py -k "spmm or shifted or config5" > $OUT/tests_dir.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests_dir.txt
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt $OUT/tests_dir.txt | tail; tail -n 4 $OUT/lap512_nodesc.txt $OUT/lap512_desc.txt $OUT/lap256_desc.txt $OUT/spmm_win.txt $OUT/spmm_dir.txt
