#!/bin/bash
# PMC passes over the SpMM kernels (tools/spmm_only.py), one small counter set per pass.
# Usage: bash tools/pmc_spmm.sh <output directory under gpurun_out> [ENV=V ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
# the assembled mesh matrix first, OUTSIDE the profiler (its generation forks workers: minutes under rocprofv3)
python - <<PY
import sys, tempfile, os
sys.path.insert(0, "$R")
from mpi_bicgstab_amd import mesh
d = os.path.join(tempfile.gettempdir(), "bicg_mesh_cache"); os.makedirs(d, exist_ok=True)
mesh.fem_unstructured(117, "generator", cache_dir=d)
PY
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  env "$@" timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p --output-format csv -- python $R/tools/spmm_only.py > $OUT/p$i.log 2>&1
done <<SETS
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
SETS
python - > $OUT/summary.txt <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/p_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_spmm' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(k[0], k[1], 'launches', len(v), 'max', max(v), 'mean', sum(v) / len(v))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +200k -delete
