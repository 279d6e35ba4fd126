#!/bin/bash
# round 4, GPU call 24: the "queue stall" named -- code objects loaded at the first launch from a translation unit (BICG_PRELOAD=0)
# against loaded by bicg_create (default); end to end at Transport size with the threaded set-up (IO / set-up / solve)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c24
mkdir -p $OUT
cd $R
BICG_PRELOAD=0 BICG_PLAN_TRACE=1 timeout 120 python tools/preload_check.py > $OUT/preload_off.txt 2>&1
BICG_PRELOAD=1 BICG_PLAN_TRACE=1 timeout 120 python tools/preload_check.py > $OUT/preload_on.txt 2>&1
grep -v "plan  [a-z]" $OUT/preload_off.txt; grep -E "code objects|PRELOAD|chunk|first" $OUT/preload_on.txt
BICG_MAX_ITER=200 REF_MAX_ITER=200 timeout 600 bash tools/end_to_end.sh > $OUT/e2e.log 2>&1
cp gpurun_out/end_to_end.txt $OUT/end_to_end.txt; cat $OUT/end_to_end.txt
