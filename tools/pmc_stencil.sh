#!/bin/bash
# Counter passes over 10 products back to back on the 512^3 Laplacian (tools/lap512_spmv.py), one small set per pass.
# Usage: bash tools/pmc_stencil.sh <output directory under gpurun_out> [ENV=V ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  env "$@" timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p --output-format csv -- python $R/tools/lap512_spmv.py > $OUT/p$i.log 2>&1
done <<SETS
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
SETS
python - > $OUT/summary.txt <<PY
import csv, glob, collections
print("settings: $*")
for f in sorted(glob.glob("$OUT/p*/**/p_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_spmv_stencil' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        v = v[2:] if len(v) > 4 else v
        print(f"  {k:36s} {sum(v)/len(v):18.1f}   ({len(v)} launches)")
for f in sorted(glob.glob("$OUT/p1/**/p_kernel_trace.csv", recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_spmv_stencil' in r['Kernel_Name']: d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in d.items(): print(f"  {k}: duration under counter pass 1: {sum(v[2:])/len(v[2:]):.1f} us ({len(v)} launches)")
PY
cat $OUT/summary.txt; grep -l -i "error\|invalid" $OUT/*.log | head
find $OUT -name "*.csv" -size +200k -delete
