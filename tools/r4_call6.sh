#!/bin/bash
# round 4, GPU call 6: the whole GPU suite WITHOUT -x (call 5 stopped at its first failure, a bench-contract assertion), the
# driver-flag bench line with fractions on the bytes the layout moves
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c6
mkdir -p $OUT
cd $R
timeout 1300 python -m pytest tests -q -m gpu --durations=15 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
grep -E "passed|failed|FAILED|ERROR|exit status" $OUT/gpu_suite.txt | tail -30; tail -22 $OUT/gpu_suite.txt
