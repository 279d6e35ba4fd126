#!/bin/bash
# rocprofv3 kernel stats of the 16-shift solvers on the Transport-shaped matrix
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
python $R/tools/ab_shift.py 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_shift -o shift --output-format csv -- python $R/tools/ab_shift.py > $OUT/prof_shift.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof_shift/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x {int(r["Calls"]):5d}  {float(r["Percentage"]):5.1f}%  {r["Name"][:150]}')
PY
