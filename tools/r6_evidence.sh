#!/bin/bash
# Round 6's evidence from ONE build (the caller names the commit): kernel statistics of the headline, the three numberings of the mesh
# matrix, the 512^3 Laplacian and the SpMM (rocprofv3 --kernel-trace --stats, one workload each), the bench line with the driver's
# flags, the GPU suite. Run on the GPU box: gpurun -- 'bash tools/r6_evidence.sh'. Outputs under gpurun_out/r06/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06
rm -rf $OUT; mkdir -p $OUT
QUIET="--steps 60 --warmup 10 --no-cpu-baseline --no-variants --no-extras --no-traffic --no-stream --regions 1"
stats() {   # name, command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o s --output-format csv -- "$@" > $OUT/$name.log 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
}
# the assembled mesh matrix first, OUTSIDE the profiler (its generation forks 32 workers: under rocprofv3 that took the 600 s limit)
python - <<PY
import sys, tempfile, os
sys.path.insert(0, "$R")
from mpi_bicgstab_amd import mesh
d = os.path.join(tempfile.gettempdir(), "bicg_mesh_cache"); os.makedirs(d, exist_ok=True)
mesh.fem_unstructured(117, "generator", cache_dir=d)
PY
stats headline_plain python $R/bench.py $QUIET
stats mesh_rcm_plain python $R/bench.py $QUIET --workload mesh --numbering rcm
stats mesh_generator_plain python $R/bench.py $QUIET --workload mesh --numbering generator
stats mesh_random_plain python $R/bench.py $QUIET --workload mesh --numbering random
stats laplace512 python $R/tools/lap512_only.py
stats spmm python $R/tools/spmm_only.py
stats spmm_mesh_rcm env SPMM_MATRIX=mesh_rcm python $R/tools/spmm_only.py
stats spmm_fem_like env SPMM_MATRIX=fem_like python $R/tools/spmm_only.py
# counters (their own passes, kernel trace only): what the SpMM and the 512^3 product move
pmc() {   # name, counter set, command...
  local name=$1 set=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$name -o p --output-format csv -- "$@" > /dev/null 2>&1
  python - "$name" "$set" $(find /tmp/pmc_$name -name "p_counter_collection.csv" | head -1) >> $OUT/counters.txt <<PY
import csv, sys, collections
name, cset, path = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if "k_spmm" in k or "k_spmv_stencil" in k:
        acc[(k.split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, cn), v in sorted(acc.items()):
    v = v[2:] if len(v) > 4 else v
    print(f"{name:10s} {k:62s} {cn:24s} mean per launch {sum(v)/len(v):16.1f}  ({len(v)} launches)")
PY
}
: > $OUT/counters.txt
for set in FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  pmc spmm "$set" python $R/tools/spmm_only.py
  pmc laplace512 "$set" python $R/tools/lap512_spmv.py
  pmc spmm_rcm "$set" env SPMM_MATRIX=mesh_rcm python $R/tools/spmm_only.py
done
cd $R
s=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
echo "bench.py wall seconds: $(( $(date +%s) - s ))" >> $OUT/bench_stderr.txt
cp bench_full.json $OUT/bench_full.json
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -12 > $OUT/gpu_suite.txt
wc -c $OUT/bench_line.json; tail -2 $OUT/bench_stderr.txt; cat $OUT/gpu_suite.txt
cat $OUT/counters.txt
for f in $OUT/*_kernel_stats.csv; do echo "== $f"; head -6 $f | cut -c1-150; done
