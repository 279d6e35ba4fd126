#!/bin/bash
# round 4, GPU call 5 (first of the second session): the whole GPU suite with durations, the driver-flag bench line, rocprofv3
# kernel stats of the headline with the round's new defaults (alternating, XCD-contiguous products)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c5
mkdir -p $OUT
cd $R
timeout 1100 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
cd /tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o $name --output-format csv -- "$@" > $OUT/prof_$name.log 2>&1
  cp $(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/prof_$name
}
B="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants --no-extras --no-traffic --no-stream"
prof headline_plain $B
prof headline_ca $B --method ca_bicgstab
prof headline_pipe $B --method pipe_bicgstab
cd $R
tail -25 $OUT/gpu_suite.txt; cat $OUT/bench_driver_flags.json | head -c 6000; tail -3 $OUT/bench_driver_flags.err; head -8 $OUT/headline_plain_kernel_stats.csv
