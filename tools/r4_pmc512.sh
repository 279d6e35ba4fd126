#!/bin/bash
# round 4: what bounds the product of the 512^3 Laplacian once it streams no matrix? PMC passes over 10 products back to back
# (one small counter set each), plane-block order of the groups and natural order. Output: gpurun_out/r4pmc512/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4pmc512
rm -rf $OUT; mkdir -p $OUT
cfg=0
for env in ${PMC512_ENVS:-"BICG_SELL_BLOCK=128" "BICG_SELL_BLOCK=0"}; do
  cfg=$((cfg+1)); i=0
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    env $env timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/c${cfg}_p$i -o p --output-format csv -- python $R/tools/lap512_spmv.py > $OUT/c${cfg}_p$i.log 2>&1
  done <<SETS
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD
FETCH_SIZE
WRITE_SIZE
SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM
TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum
SETS
done
python - > $OUT/summary.txt <<PY
import csv, glob, collections
names = {1: "first setting of PMC512_ENVS (default: plane-block order, B = 128)", 2: "second setting (default: natural order)"}
for cfg in (1, 2):
    print("==", names[cfg])
    for f in sorted(glob.glob("$OUT/c%d_p*/**/p_counter_collection.csv" % cfg, recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'k_spmv_sell<0' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            v = v[3:] if len(v) > 6 else v
            print(f"  {k:42s} {sum(v)/len(v):18.1f}   ({len(v)} launches)")
    for f in sorted(glob.glob("$OUT/c%d_p1/**/p_kernel_trace.csv" % cfg, recursive=True)):
        d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'k_spmv_sell<0' in r['Kernel_Name']]
        if d: print(f"  kernel duration under the counter pass 1: {sum(d[3:])/len(d[3:]):.1f} us")
PY
cat $OUT/summary.txt; grep -l -i "error\|invalid" $OUT/*.log | head
