#!/bin/bash
# round 4, GPU call 27 (evidence run of the round's final build): the whole GPU suite, the driver-flag bench line, rocprofv3 kernel
# stats of the headline (plain / CA / pipelined), of the 512^3 leg, of the SpMM and of the shifted solvers on a rank of 8
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c27
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
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
prof lap512 python $R/tools/lap512_only.py
prof spmm python $R/tools/spmm_only.py
cd $R
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/gpu_suite.txt | tail -20; head -6 $OUT/headline_plain_kernel_stats.csv | cut -c1-160; head -8 $OUT/lap512_kernel_stats.csv | cut -c1-160; tail -2 $OUT/bench_driver_flags.err; cut -c1-300 $OUT/bench_driver_flags.json
