#!/bin/bash
# round 3, call A: parity of the bench workloads at bench size, STREAM numbers, rows-over-lanes on banded b = 512
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
BICG_STREAM_VERBOSE=1 python - > gpurun_out/stream.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
from mpi_bicgstab_amd import hipsolver as H
H.lib().bicg_comm_init_single(0)
for gb in (1, 2):
    for kind in ("copy", "triad", "read8", "read16"):
        best = max(H.stream_bench(kind, gb << 30, 20) for _ in range(3))
        print(f"{kind:7s} {gb} GiB per array: {best:8.1f} GB/s", flush=True)
PY
cat gpurun_out/stream.txt
B="python bench.py --workload banded --half-bandwidth 512 --steps 100 --warmup 10 --no-cpu-baseline --no-variants --no-extras --no-traffic"
for rs in 1 0; do
  BICG_ROWSPLIT=$rs timeout 200 $B 2>gpurun_out/b512_rs$rs.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rowsplit=$rs b512 plain', '%.1f us' % (1e3*d['value']), 'spmv in-solver %.1f us' % (1e3*d['roofline']['avg_launch_ms']), 'b2b %.1f us' % (1e3*d['roofline']['back_to_back_spmv_ms']), 'frac', round(d['roofline']['frac'],3))" | tee -a gpurun_out/b512.txt
done
B64="python bench.py --workload banded --half-bandwidth 64 --steps 100 --warmup 10 --no-cpu-baseline --no-variants --no-extras --no-traffic"
for rs in 1 0; do
  BICG_ROWSPLIT=$rs timeout 200 $B64 2>gpurun_out/b64_rs$rs.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rowsplit=$rs b64 plain', '%.1f us' % (1e3*d['value']), 'spmv in-solver %.1f us' % (1e3*d['roofline']['avg_launch_ms']), 'b2b %.1f us' % (1e3*d['roofline']['back_to_back_spmv_ms']), 'frac', round(d['roofline']['frac'],3))" | tee -a gpurun_out/b512.txt
done
timeout 900 python -m pytest tests/test_bench_workloads.py -x -q -m gpu --durations=10 2>&1 | tail -25 | tee gpurun_out/test_bench_workloads.log
