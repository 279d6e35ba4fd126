#!/bin/bash
# round 4, GPU call 10: constant slices (512^3 Laplacian without a matrix stream in the interior), windowed SpMM with the row head
# in registers, the whole GPU suite, bench line, per-kernel stats of the 512^3 leg and of the SpMM
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c10
mkdir -p $OUT
cd $R
timeout 200 python tools/spmm_only.py > $OUT/spmm_only.txt 2>&1
timeout 300 python tools/lap512_only.py > $OUT/lap512.txt 2>&1
BICG_SELL_CONSTANT=0 timeout 300 python tools/lap512_only.py > $OUT/lap512_no_constant.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
cd /tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o $name --output-format csv -- "$@" > $OUT/prof_$name.log 2>&1
  cp $(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/prof_$name
}
prof spmm python $R/tools/spmm_only.py
prof lap512 python $R/tools/lap512_only.py
cd $R
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/gpu_suite.txt | tail -20
cat $OUT/spmm_only.txt $OUT/lap512.txt $OUT/lap512_no_constant.txt; head -4 $OUT/spmm_kernel_stats.csv; head -12 $OUT/lap512_kernel_stats.csv; tail -4 $OUT/bench_driver_flags.err
