#!/bin/bash
# Ragged rows across ranks: the launch with the halo exchange inside (k_spmv_sell's loop over every group) against separate
# launches (interior through k_spmv_jagd / k_spmv_jagw). Two ranks SHARING the GPU hold 800 k rows each of the mesh matrix.
R=${GRAFT_REPO_ROOT:-/root/repo}
for numbering in rcm generator; do
  for fused in 1 0; do
    for method in bicgstab pipe_bicgstab; do
      BICG_PLAN="halo-fused=$fused" python $R/bench.py --gpus 2 --workload mesh --numbering $numbering --method $method --transport host-p2p \
        --no-extras --no-variants --no-cpu-baseline --no-stream --no-rccl-leg --steps 100 --warmup 20 > /tmp/line.json 2> /tmp/err.txt
      python - <<PY
import json
d = json.load(open("$R/bench_full.json"))
print("numbering $numbering halo-fused=$fused $method: %.4f ms/iteration, flags %s, product %.1f us (events), genuine %s" % (d["value"], [f for f in d["config"]["flags"] if f in ("ll_fused", "jagged", "window", "p2p")], 1e3 * d["roofline"]["avg_launch_ms"], d["config"]["iterations_genuine"]))
PY
    done
  done
done
