#!/bin/bash
# round 4, GPU call 28: the stalled first region of pipe_bicgstab (65 ms in call 27's bench line) outside bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c28
mkdir -p $OUT
cd $R
timeout 120 python tools/stall_check.py > $OUT/stall_regions.txt 2>&1
STALL_SINGLE_STEPS=1 timeout 120 python tools/stall_check.py > $OUT/stall_single_steps.txt 2>&1
AMD_LOG_LEVEL=3 timeout 120 python tools/stall_check.py > $OUT/stall_regions_log3.txt 2>&1
cat $OUT/stall_regions.txt $OUT/stall_single_steps.txt
grep -c "" $OUT/stall_regions_log3.txt
timeout 200 python -m pytest -q -m gpu tests/test_bench_torchrun.py -k single_gpu_line > $OUT/test_line.txt 2>&1; tail -3 $OUT/test_line.txt
