#!/bin/bash
# round 4, GPU call 9: windowed SpMM with batched staging, shifted solvers (both) as persistent launches with the pass over the
# shifts behind the hand-off, ranks without rows, the tests that failed in call 8
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c9
mkdir -p $OUT
cd $R
timeout 200 python tools/spmm_only.py > $OUT/spmm_only.txt 2>&1
timeout 200 python tools/shifted_small_rank.py > $OUT/shifted_small_rank.txt 2>&1
timeout 1500 python -m pytest -q --durations=10 -m gpu \
  "tests/test_bench_torchrun.py::test_bench_single_gpu_line_has_every_leg" \
  "tests/test_bench_workloads.py::test_shifted_on_a_rank_of_8_is_one_persistent_launch" \
  "tests/test_bench_workloads.py::test_fem_like_as_benchmarked" \
  "tests/test_bench_workloads.py::test_config5_as_benchmarked" \
  "tests/test_dropin_host.py::test_reference_shifted_driver_linked_against_hip_library" \
  "tests/test_full_size.py::test_spmm_16_vectors_reads_the_matrix_once" \
  tests/test_shifted.py tests/test_switching.py tests/test_multirank.py tests/test_multirank_fullsize.py > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_spmm -o spmm --output-format csv -- python $R/tools/spmm_only.py > $OUT/prof_spmm.log 2>&1
cp $(find $OUT/prof_spmm -name "*kernel_stats.csv" | head -1) $OUT/spmm_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof_spmm
cd $R
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt | tail -20
cat $OUT/shifted_small_rank.txt $OUT/spmm_only.txt; head -5 $OUT/spmm_kernel_stats.csv
