#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the bench workload.
# Outputs under gpurun_out/prof_*; tools/summarize_profile.py turns them into profiles/*.
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
BENCH="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats --output-format csv -- $BENCH > $OUT/prof_stats.log 2>&1
# PMC passes, each alone (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2)
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch --output-format csv -- $BENCH > $OUT/prof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o write --output-format csv -- $BENCH > $OUT/prof_write.log 2>&1
ls -R $OUT | head -50
