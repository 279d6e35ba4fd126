#!/bin/bash
# One GPU call to start a round with: the whole GPU suite with durations (and its exit status -- see profiles/NOTES.md,
# "Exit status 134"), the driver-flag bench line, the small-rank table, the cost of a persistent launch, and two things
# that were added after round 3's GPU budget was spent and have only been run by the driver since:
#   tests/test_shifted.py::test_hip_shifted_dropin_symbols, oracle/_ref/shifted_dropin (the reference's main_shifted.c
#   linked against the library) on a small Matrix-Market file next to the all-reference build.
# Usage: /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/next_round_first_call.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/first_call
mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
timeout 300 bash tools/small_rank_times.sh > $OUT/small_rank_times.txt 2>&1
timeout 120 python tools/persist_chunk_times.py > $OUT/persist_chunk_times.txt 2>&1
python - > $OUT/shifted_dropin.txt 2>&1 <<'PY'
import os, subprocess, sys
sys.path.insert(0, '.')
from mpi_bicgstab_amd import synth
A = synth.from_offsets(3001, (0, 1, -1, 60, -60), diag_base=9.0, seed=3)
row, col, val = synth.colmajor_coo(A)
with open('/tmp/sh.mtx', 'w') as f:
    f.write('%%MatrixMarket matrix coordinate real general\n%d %d %d\n' % (A.rows, A.cols, A.nnz))
    for i, j, v in zip(row.tolist(), col.tolist(), val.tolist()):
        f.write('%d %d %r\n' % (i + 1, j + 1, v))
for exe in ('oracle/_ref/shifted_dropin',):
    for np_ in (1, 2):
        out = subprocess.run(['/opt/conda/bin/mpiexec', '-n', str(np_), exe, '/tmp/sh.mtx'], capture_output=True, text=True, timeout=200,
                             env=dict(os.environ, BICG_TRANSPORT='host'))
        print('==', exe, 'P =', np_, 'rc', out.returncode)
        print(out.stdout[-1500:])
        print(out.stderr[-500:])
PY
tail -5 $OUT/gpu_suite.txt
