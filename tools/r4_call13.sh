#!/bin/bash
# round 4, GPU call 13: windowed SpMM with batched LDS reads, headline after the tail-finish change
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c13
mkdir -p $OUT
cd $R
timeout 200 python tools/spmm_only.py > $OUT/spmm_only.txt 2>&1
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab timeout 300 python tools/ab.py "" > $OUT/ab_default.txt 2>&1
timeout 600 python -m pytest -q -s -m gpu tests/test_shifted.py "tests/test_full_size.py::test_spmm_16_vectors_reads_the_matrix_once" "tests/test_bench_workloads.py::test_fem_like_as_benchmarked" tests/test_multirank.py > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
grep -E "passed|failed|^FAILED|^ERROR|exit status|SpMM 16|fem_like: SpMM" $OUT/tests.txt | tail; cat $OUT/spmm_only.txt $OUT/ab_default.txt
