#!/bin/bash
# k_spmm_jpipe (csrc/bicg_spmm_jag.hip) on the ragged matrices: 16 vectors against k_spmm_win (bit for bit), then the kernel with parts
# switched off (tools/spmm_jag_skip.py). Writes gpurun_out/spmm_jag_probe.txt
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
out=gpurun_out/spmm_jag_probe.txt; : > $out
for m in ${MATRICES:-fem_like mesh_generator mesh_rcm}; do
  for t in ${VARIANTS:-none}; do
    [ "$t" = none ] && t=""
    echo "== $m BICG_TEST=$t" >> $out
    SPMM_MATRIX=$m BICG_TEST="$t" timeout 600 python tools/spmm_only.py >> $out 2>&1
  done
done
for m in ${SKIP_MATRICES:-mesh_rcm}; do
SPMM_MATRIX=$m SPMM_TOKENS="${SKIPS:-spmm-skip=0;spmm-skip=1;spmm-skip=2;spmm-skip=16;spmm-skip=8;spmm-skip=4;spmm-skip=32;spmm-skip=64;spmm-skip=127}" timeout 300 python tools/spmm_jag_skip.py >> $out 2>&1
done
cat $out
