#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c7
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
tail -22 $OUT/gpu_suite.txt
