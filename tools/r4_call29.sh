#!/bin/bash
# round 4, GPU call 29: the same sequence with torch's runtime in the process (as in bench.py), three processes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4c29
mkdir -p $OUT
cd $R
for i in 1 2 3; do STALL_TORCH=1 timeout 120 python tools/stall_check.py > $OUT/stall_torch_$i.txt 2>&1; done
grep -h "try" $OUT/stall_torch_*.txt | awk '{print $1, $2, $3, $10}' | sort | uniq -c | sort -k1,1nr | head -40
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic --no-stream > $OUT/bench_variants.json 2> $OUT/bench_variants.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4c29/bench_variants.json") if l.startswith("{")][0])
for k,v in d["timed_regions_ms"].items(): print(k[-40:], v)
PY
