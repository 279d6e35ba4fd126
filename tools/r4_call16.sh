#!/bin/bash
# round 4, GPU call 16: the wavefront's number as a scalar in the sliced-ELL row product (slice metadata through the scalar cache,
# scalar branches): headline, 512^3, parity
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c16
mkdir -p $OUT
cd $R
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab timeout 300 python tools/ab.py "" > $OUT/ab_default.txt 2>&1
timeout 300 python tools/lap512_only.py > $OUT/lap512.txt 2>&1
timeout 900 python -m pytest -q -m gpu tests/test_gpu_parity.py tests/test_full_size.py tests/test_bench_workloads.py > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt | tail; cat $OUT/ab_default.txt $OUT/lap512.txt
