#!/bin/bash
# round 4, GPU call 31: the four wavefronts of a workgroup on four consecutive grid lines WITH their own rows exchanged through LDS
# (BICG_SELL_YGROUP=1): parity of the Laplacian tests under it, 512^3 / 256^3 against the default
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c31
mkdir -p $OUT
cd $R
BICG_SELL_YGROUP=1 timeout 400 python -m pytest -q -m gpu -x tests/test_full_size.py -k "list_driven or device_side or slab_generator" > $OUT/tests_ygroup.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests_ygroup.txt
tail -4 $OUT/tests_ygroup.txt
timeout 100 python tools/lap512_only.py > $OUT/lap512_default.txt 2>&1
BICG_SELL_YGROUP=1 timeout 100 python tools/lap512_only.py > $OUT/lap512_ygroup_lds.txt 2>&1
BICG_SELL_YGROUP=1 timeout 100 python tools/lap512_only.py 256 > $OUT/lap256_ygroup_lds.txt 2>&1
tail -n 3 $OUT/lap512_default.txt $OUT/lap512_ygroup_lds.txt $OUT/lap256_ygroup_lds.txt
