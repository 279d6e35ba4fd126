#!/bin/bash
# round 4, GPU call 26: are the 512^3 vectors (exactly 1 GiB apart) in each other's way? distance between vectors padded
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c26
mkdir -p $OUT
cd $R
for pad in 0 544 4128 33824 1048608; do
  BICG_STRIDE_PAD=$pad timeout 200 python tools/lap512_only.py > $OUT/lap512_pad$pad.txt 2>&1
  echo "pad $pad: $(grep -E 'bicgstab|spmv' $OUT/lap512_pad$pad.txt | tr '\n' ' ')"
done
