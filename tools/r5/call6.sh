#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c6
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_bench_torchrun.py tests/test_multirank.py tests/test_shifted.py -x -q 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt
