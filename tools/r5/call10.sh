#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c10
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 560 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
tail -20 $OUT/gpu_suite.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
tail -12 $OUT/bench_driver_flags.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats_head -o s --output-format csv -- python $R/bench.py --inner --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic > $OUT/stats_head.log 2>&1
cp $(find $OUT/stats_head -name "s_kernel_stats.csv" | head -1) $OUT/headline_plain_kernel_stats.csv 2>/dev/null
head -8 $OUT/headline_plain_kernel_stats.csv | cut -c1-200
