#!/bin/bash
# round 4, GPU call 20: order of the 256-row groups of the 512^3 Laplacian after the descriptor change -- in-plane block size B
# (BICG_SELL_BLOCK; 1 = the same in-plane group through all planes of an XCD's share) x groups per workgroup; the fastest
# setting then runs the solvers
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c20
mkdir -p $OUT
cd $R
: > $OUT/sweep.txt
for B in 0 1 4 32 256; do
  for G in 4 8 16; do
    [ $B = 0 ] && [ $G != 8 ] && continue
    r=$(BICG_SELL_BLOCK=$B BICG_SELL_GPW=$G BICG_SELL_GPW_DOTS=$G timeout 100 python tools/lap512_spmv.py 2>&1 | tail -1)
    echo "B=$B GPW=$G  $r" >> $OUT/sweep.txt
  done
done
cat $OUT/sweep.txt
best=$(awk '{print $NF, $1, $2}' $OUT/sweep.txt | sort -g | head -1)
bB=$(echo $best | sed 's/.*B=\([0-9]*\).*/\1/'); bG=$(echo $best | sed 's/.*GPW=\([0-9]*\).*/\1/')
echo "best: $best -> B=$bB GPW=$bG" | tee -a $OUT/sweep.txt
BICG_SELL_BLOCK=$bB BICG_SELL_GPW=$bG BICG_SELL_GPW_DOTS=$bG timeout 200 python tools/lap512_only.py > $OUT/lap512_best.txt 2>&1
tail -n 4 $OUT/lap512_best.txt
