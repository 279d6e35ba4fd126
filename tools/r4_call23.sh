#!/bin/bash
# round 4, GPU call 23: set-up on several host threads (seconds per part, 1 / 8 / 32 threads), windowed SpMM with 4 vectors per
# window, then the whole GPU suite and the driver-flag bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c23
mkdir -p $OUT
cd $R
for nt in 1 8 32; do
BICG_PLAN_THREADS=$nt BICG_PLAN_TRACE=1 timeout 300 python - > $OUT/plan_trace_$nt.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
for name, A in (("transport_like", synth.transport_like(scale_decades=2.0)), ("fem_like", synth.fem_like(scale_decades=2.0))):
    blocks = H.single_rank_blocks(A)
    for rep in range(2):
        t = time.perf_counter()
        ctx = H.Context(blocks)
        print(name, "bicg_create %.4f s" % (time.perf_counter() - t), flush=True)
        ctx.close()
PY
done
grep -h "bicg_create" $OUT/plan_trace_1.txt $OUT/plan_trace_8.txt $OUT/plan_trace_32.txt
BICG_SPMM_NV=4 timeout 100 python tools/spmm_only.py > $OUT/spmm_nv4.txt 2>&1; tail -1 $OUT/spmm_nv4.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
grep -E "passed|failed|^FAILED|^ERROR|exit status" $OUT/gpu_suite.txt | tail -20; tail -3 $OUT/bench_driver_flags.err; cut -c1-400 $OUT/bench_driver_flags.json
