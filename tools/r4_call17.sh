#!/bin/bash
# round 4, GPU call 17: row-pointer loads skipped for list-driven slices; then the evidence set -- the whole GPU suite, the
# driver-flag bench line, rocprofv3 kernel stats of the headline (plain / CA / pipelined), of the 512^3 leg and of the SpMM
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c17
mkdir -p $OUT
cd $R
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab timeout 300 python tools/ab.py "" > $OUT/ab_default.txt 2>&1
timeout 300 python tools/lap512_only.py > $OUT/lap512.txt 2>&1
timeout 200 python tools/spmm_only.py > $OUT/spmm_only.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
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
prof shifted_small_rank python $R/tools/shifted_small_rank.py
cd $R
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/gpu_suite.txt | tail -20
cat $OUT/ab_default.txt $OUT/lap512.txt $OUT/spmm_only.txt; head -8 $OUT/headline_plain_kernel_stats.csv; head -12 $OUT/lap512_kernel_stats.csv; tail -3 $OUT/bench_driver_flags.err
