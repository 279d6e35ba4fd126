#!/bin/bash
# round 4, GPU call 12: headline after moving the constant-slice code into layouts of its own (the run-time branch had cost the
# dot-carrying products 11 us each), 512^3 with those layouts
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c12
mkdir -p $OUT
cd $R
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab timeout 300 python tools/ab.py "" > $OUT/ab_default.txt 2>&1
BICG_SELL_BLOCK=128 timeout 300 python tools/lap512_only.py > $OUT/lap512_block128.txt 2>&1
timeout 300 python tools/lap512_only.py > $OUT/lap512.txt 2>&1
timeout 600 python -m pytest -q -m gpu tests/test_gpu_parity.py "tests/test_full_size.py::test_laplace512_device_plan_at_bench_size" "tests/test_full_size.py::test_device_side_plan_matches_host_plan" > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt | tail; cat $OUT/ab_default.txt $OUT/lap512.txt $OUT/lap512_block128.txt
