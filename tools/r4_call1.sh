#!/bin/bash
# round 4, GPU call 1: the new parity tests, the alternating-direction SpMV A/B, device limits, the counter list, per-kernel
# stats of the 256^3 leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4c1
mkdir -p $OUT
python - > $OUT/devprops.txt 2>&1 <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(p)
for k in dir(p):
    if not k.startswith('_'):
        try: print(k, getattr(p, k))
        except Exception as e: print(k, 'ERR', e)
PY
timeout 900 python -m pytest -x -q --durations=10 \
  "tests/test_full_size.py::test_device_side_plan_matches_host_plan" \
  "tests/test_full_size.py::test_laplace512_device_plan_at_bench_size" \
  "tests/test_bench_workloads.py::test_transport_rank_of_8_as_benchmarked" \
  "tests/test_multirank_fullsize.py::test_two_small_ranks_persistent_with_halo" \
  "tests/test_multirank_fullsize.py::test_fullsize_partition_against_oracle[host-p2p-8]" \
  "tests/test_gpu_parity.py" > $OUT/tests.txt 2>&1; echo "pytest exit status $?" >> $OUT/tests.txt
AB_METHODS=bicgstab,ca_bicgstab,pipe_bicgstab timeout 400 python tools/ab.py "" "BICG_SELL_ALT=1" "BICG_SELL_ALT=1;BICG_SELL_NT=0" "BICG_SELL_ALT=1;BICG_SELL_XCD=1" > $OUT/ab_alt.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --list-avail > $OLDPWD/$OUT/counters_avail.txt 2>&1)
tail -5 $OUT/tests.txt; cat $OUT/ab_alt.txt
