#!/bin/bash
# round 4, GPU call 7: the whole GPU suite (no -x), the driver-flag bench line, 16 shifts on a 200 k-row rank as persistent
# launches against the multi-launch forms, pipelined headline with / without the fused two-launch form
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c7
mkdir -p $OUT
cd $R
timeout 1300 python -m pytest tests -q -m gpu --durations=15 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
timeout 300 python tools/shifted_small_rank.py > $OUT/shifted_small_rank.txt 2>&1
AB_REPS=3 AB_METHODS=pipe_bicgstab,pipe_bicgstab_rr timeout 300 python tools/ab.py "" "BICG_FUSE_PIPE=0" "BICG_FUSE_PIPE=1" > $OUT/ab_fuse_pipe.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/gpu_suite.txt | tail -30
cat $OUT/shifted_small_rank.txt $OUT/ab_fuse_pipe.txt; tail -4 $OUT/bench_driver_flags.err
