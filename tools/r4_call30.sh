#!/bin/bash
# round 4, GPU call 30: the FEM-like matrix (jagged slices + x windows) under the knobs of this round
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c30
mkdir -p $OUT
cd $R
AB_MATRIX=fem_like AB_REPS=3 AB_METHODS=bicgstab,pipe_bicgstab timeout 400 python tools/ab.py "" "BICG_SELL_WINDOW=0" "BICG_SELL_SORT=0" "BICG_SELL_ALT=0" "BICG_SELL_XCD=0" "BICG_SELL_GPW=2;BICG_SELL_GPW_DOTS=2" "BICG_SELL_GPW=4;BICG_SELL_GPW_DOTS=4" > $OUT/ab_fem.txt 2>&1
cat $OUT/ab_fem.txt
