#!/bin/bash
out=gpurun_out/${1:-final}; mkdir -p $out
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "ms/iter", round(d["value"],4), "spmv us", round(1e3*d["roofline"]["avg_launch_ms"],1), "genuine", d["config"].get("iterations_genuine"), d["config"].get("relres_after_timed_region"))'
timeout 200 $B --workload fem_like 2>/dev/null | python -c "$show" "fem_like plain"
timeout 200 $B --workload fem_like --method pipe_bicgstab 2>/dev/null | python -c "$show" "fem_like pipe"
timeout 300 python -m pytest tests/test_bench_torchrun.py tests/test_bench_contract.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tail -3
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
for k,e in d["extras"].items():
    print(k, {m: (round(e[m]["ms_per_iteration"],4), round(e[m]["frac"],3), e[m].get("iterations_genuine")) for m in e if isinstance(e[m], dict) and "ms_per_iteration" in e[m]}, "spmv", round(e["spmv_back_to_back"]["ms"]*1e3,1))
PY
