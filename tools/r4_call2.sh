#!/bin/bash
# round 4, GPU call 2: whole GPU suite on the rewritten persistent kernels, the direction / placement / streaming factorial,
# persistent forms at 200 k / 400 k / 800 k rows
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4c2
mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu --maxfail=6 --durations=12 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab timeout 600 python tools/ab.py "" "BICG_SELL_ALT=1" "BICG_SELL_XCD=1" "BICG_SELL_ALT=1;BICG_SELL_XCD=1" \
   "BICG_SELL_NT=0" "BICG_SELL_ALT=1;BICG_SELL_NT=0" "BICG_SELL_XCD=1;BICG_SELL_NT=0" "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_SELL_NT=0" \
   "BICG_SELL_NT=1" "BICG_SELL_ALT=1;BICG_SELL_NT=1" "BICG_SELL_XCD=1;BICG_SELL_NT=1" "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_SELL_NT=1" > $OUT/ab_factorial.txt 2>&1
AB_REPS=3 AB_METHODS=bicgstab,ca_bicgstab timeout 400 python tools/ab.py "BICG_SELL_ALT=1;BICG_SELL_XCD=1" "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_VEC_REV=1" "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_VEC_REV=2" \
   "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_VEC_REV=4" "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_VEC_REV=3" "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_VEC_REV=5" \
   "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_VEC_REV=6" "BICG_SELL_ALT=1;BICG_SELL_XCD=1;BICG_VEC_REV=7" > $OUT/ab_vecrev.txt 2>&1
for spec in "200264 pipe_bicgstab" "200264 pipe_bicgstab_rr 5 2" "400528 pipe_bicgstab" "801056 pipe_bicgstab"; do
  echo "=== $spec" >> $OUT/persist_check.txt
  BICG_DEBUG=1 timeout 200 python tools/persist_check.py $spec >> $OUT/persist_check.txt 2>&1
done
tail -15 $OUT/gpu_suite.txt; cat $OUT/ab_factorial.txt $OUT/ab_vecrev.txt; grep -v "^bicgstab_hip: rank 0: context\|HIP error" $OUT/persist_check.txt | tail -60
