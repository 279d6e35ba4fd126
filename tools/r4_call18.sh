#!/bin/bash
# round 4, GPU call 18: products of the pipelined iteration without the wait for `done` in workgroups that have no part in the open
# dot group; parity of the pipelined solvers (single GPU, multi-rank, full size)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c18
mkdir -p $OUT
cd $R
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab,pipe_bicgstab_rr timeout 300 python tools/ab.py "" > $OUT/ab_default.txt 2>&1
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_parity.py tests/test_full_size.py tests/test_multirank.py tests/test_multirank_fullsize.py tests/test_comm_path_one_gpu.py > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/tests.txt | tail; cat $OUT/ab_default.txt
