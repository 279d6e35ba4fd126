#!/bin/bash
# End to end at Transport size: the Transport-shaped synthetic as a Matrix-Market file (~850 MB,
# 23.9 M entries), through our C host (read-once loader, device plan, resident solve) and through the reference itself
# (oracle/_ref/solver_ref_env: its loader fscanf()s the file twice per rank, src/matrix.c:268-396), at P = 1 and P = 8.
# Ranks share the one GPU of the box (host-staged MPI transport) -- the IO / set-up split is what this is about.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/end_to_end.txt
MTX=/tmp/transport_like.mtx
: > $OUT
python - <<'PY' 2>&1 | tee -a $OUT
import sys, time
sys.path.insert(0, '.')
import numpy as np, pandas as pd
from mpi_bicgstab_amd import synth
t = time.time()
A = synth.transport_like(scale_decades=2.0)
row, col, val = synth.colmajor_coo(A)
with open('/tmp/transport_like.mtx', 'w') as f:
    f.write('%%MatrixMarket matrix coordinate real general\n')
    f.write(f'{A.rows} {A.cols} {A.nnz}\n')
pd.DataFrame({'i': row + 1, 'j': col + 1, 'v': val}).to_csv('/tmp/transport_like.mtx', sep=' ', header=False, index=False, mode='a', float_format='%.17g')
import os
print(f'wrote /tmp/transport_like.mtx: {os.path.getsize("/tmp/transport_like.mtx") / 1e6:.0f} MB, {A.nnz} entries, {time.time() - t:.0f} s')
PY
run() {   # label, command...
  local label="$1"; shift
  local t0=$(date +%s.%N)
  "$@" > /tmp/e2e.out 2>/tmp/e2e.err
  local rc=$? t1=$(date +%s.%N)
  echo "== $label (rc $rc, wall $(python -c "print('%.2f' % ($t1 - $t0))") s)" | tee -a $OUT
  grep -E "IO time|Setup time|Total iter|Final r|Total time|Avg time" /tmp/e2e.out | sed 's/^/   /' | tee -a $OUT
  [ $rc -ne 0 ] && tail -3 /tmp/e2e.err | tee -a $OUT
}
export BICG_MAX_ITER=200 REF_MAX_ITER=200
HOST=mpi-bicgstab_amd/host/bicg_solver_host
run "ours P=1 (bicg_solver_host, MI355X)"            $HOST $MTX bicgstab
run "ours P=1, COO->CSR on the GPU (BICG_INGEST=device)" env BICG_INGEST=device $HOST $MTX bicgstab
run "ours P=8, ranks share the GPU (MPI-staged)"     /opt/conda/bin/mpiexec -n 8 $HOST $MTX bicgstab
run "reference P=1 (solver_ref_env, CPU)"            oracle/_ref/solver_ref_env $MTX bicgstab
run "reference P=8 (solver_ref_env, CPU)"            /opt/conda/bin/mpiexec -n 8 oracle/_ref/solver_ref_env $MTX bicgstab
