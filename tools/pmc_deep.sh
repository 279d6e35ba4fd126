#!/bin/bash
# Deep counter collection for the SpMV kernel alone (tools/spmv_only.py), one small --pmc set per pass.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcd
rm -rf $OUT; mkdir -p $OUT
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p --output-format csv -- python $R/tools/spmv_only.py > $OUT/p$i.log 2>&1
done <<SETS
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_avr
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum
GRBM_GUI_ACTIVE GRBM_COUNT
SETS
python - <<PY
import csv, glob, collections, os
for f in sorted(glob.glob("$OUT/*/p_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_spmv' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(f"{k:45s} {sum(v)/len(v):16.1f}")
PY
grep -l -i "error\|invalid" $OUT/*.log | head
