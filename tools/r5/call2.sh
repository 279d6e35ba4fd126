#!/bin/bash
# round 5, second GPU call: tile order / store policy of the plane-marching product, the whole GPU suite, the driver-flag bench line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c2
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 300 python tools/stencil_sweep.py 512 "" "BICG_STENCIL_XCD=0" "BICG_STENCIL_NT=1" "BICG_STENCIL_XCD=0 BICG_STENCIL_NT=1" "BICG_STENCIL_XCD=0 BICG_STENCIL_ZL=16" > $OUT/sweep512_xcd_nt.txt 2>&1
cat $OUT/sweep512_xcd_nt.txt
timeout 100 python tools/stencil_sweep.py 256 "" "BICG_STENCIL_XCD=0" "BICG_STENCIL_NT=1" "BICG_STENCIL_XCD=0 BICG_STENCIL_NT=1" > $OUT/sweep256_xcd_nt.txt 2>&1
cat $OUT/sweep256_xcd_nt.txt
timeout 900 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/gpu_suite.txt 2>&1; echo "pytest exit status $?" >> $OUT/gpu_suite.txt
tail -25 $OUT/gpu_suite.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
tail -30 $OUT/bench_driver_flags.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_driver_flags.json").read().strip().splitlines()[-1])
print("value", d["value"], "regions", d["headline_regions"])
print("roofline frac", d["roofline"]["frac"], "8d", d["roofline"]["survey_8d_frac"], "structure", d["roofline"]["structure_dependence"])
print("unstructured", d["roofline_unstructured"])
for k in ("laplace7_256_ca", "laplace7_512_ca"):
    e = d["extras"].get(k, {})
    print(k, {m: e[m]["ms_per_iteration"] for m in ("bicgstab", "ca_bicgstab") if m in e}, e.get("spmv_back_to_back", {}).get("ms"))
PY
