#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcv
mkdir -p $OUT
for v in 0 1 2; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $c | tr ' ' '_')
    BICG_SPMV_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/v${v}_$tag -o p --output-format csv -- python $R/tools/spmv_only.py > $OUT/v${v}_$tag.log 2>&1
  done
done
python - <<PY
import csv, glob, collections, os
for f in sorted(glob.glob("$OUT/*/p_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_spmv' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(os.path.dirname(f)), {k: round(sum(v)/len(v),1) for k,v in acc.items()})
PY
