#!/bin/bash
# PMC passes over 20 products back to back (tools/spmv_only.py; SPMV_KIND=fem_like for the FEM-like matrix), one small counter
# set per pass. Usage: bash tools/pmc_spmv.sh <output directory under gpurun_out> [ENV=V ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  env "$@" timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p --output-format csv -- python $R/tools/spmv_only.py > $OUT/p$i.log 2>&1
done <<SETS
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD
FETCH_SIZE
WRITE_SIZE
SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SETS
python - > $OUT/summary.txt <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/p_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_spmv' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:70], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items():
        v = v[3:] if len(v) > 6 else v
        print(f"  {k[0]:72s} {k[1]:36s} {sum(v)/len(v):18.1f}   ({len(v)} launches)")
for f in sorted(glob.glob("$OUT/p1/**/p_kernel_trace.csv", recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_spmv' in r['Kernel_Name']: d[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in d.items(): print(f"  {k}: kernel duration under counter pass 1: {sum(v[3:])/len(v[3:]):.1f} us ({len(v)} launches)")
PY
cat $OUT/summary.txt; grep -l -i "error\|invalid" $OUT/*.log | head
