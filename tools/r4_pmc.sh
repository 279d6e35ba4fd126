#!/bin/bash
# round 4: what does the in-solver sliced-ELL product wait for? PMC passes (one small counter set each, kernel trace only)
# over a 30-iteration plain BiCGStab for the group orders: default (round robin over the XCDs, always forward) and
# XCD-contiguous + alternating direction. Output: gpurun_out/r4pmc/summary.txt (per-launch averages of k_spmv_sell<1|2,...>)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4pmc
rm -rf $OUT; mkdir -p $OUT
cfg=0
for env in "BICG_SELL_ALT=0 BICG_SELL_XCD=0" "BICG_SELL_ALT=0 BICG_SELL_XCD=1" "BICG_SELL_ALT=1 BICG_SELL_XCD=0" "BICG_SELL_ALT=1 BICG_SELL_XCD=1"; do
  cfg=$((cfg+1)); i=0
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    env $env timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/c${cfg}_p$i -o p --output-format csv -- python $R/tools/solve_only.py > $OUT/c${cfg}_p$i.log 2>&1
  done <<SETS
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr TCC_CYCLE_sum TCC_EA0_WRREQ_sum
FETCH_SIZE
WRITE_SIZE
SETS
done
python - > $OUT/summary.txt <<PY
import csv, glob, collections
names = {1: "forward, round robin over XCDs (round 3)", 2: "forward, XCD-contiguous", 3: "alternating, round robin", 4: "alternating, XCD-contiguous"}
for cfg in (1, 2, 3, 4):
    print("==", names[cfg])
    for f in sorted(glob.glob("$OUT/c%d_p*/**/p_counter_collection.csv" % cfg, recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name']
            if 'k_spmv_sell<1' in kn or 'k_spmv_sell<2' in kn:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            v = v[4:] if len(v) > 8 else v
            print(f"  {k:42s} {sum(v)/len(v):18.1f}   ({len(v)} launches)")
    for f in sorted(glob.glob("$OUT/c%d_p1/**/p_kernel_trace.csv" % cfg, recursive=True)):
        d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'k_spmv_sell<1' in r['Kernel_Name'] or 'k_spmv_sell<2' in r['Kernel_Name']]
        if d: print(f"  kernel duration under the counter pass 1: {sum(d[4:])/len(d[4:]):.1f} us")
PY
cat $OUT/summary.txt; grep -l -i "error\|invalid" $OUT/*.log | head
