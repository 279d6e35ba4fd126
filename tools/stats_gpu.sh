#!/bin/bash
# kernel-trace stats of a short bench run (cheap); prints the per-kernel table
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/stats_$1
rm -rf $OUT; mkdir -p $OUT
shift
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/s_kernel_stats.csv")):
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):6.2f}%")
PY
grep -o '"value": [0-9.]*' $OUT/log.txt | head -1
