#!/bin/bash
# One gpurun call: the COMPLETE GPU test suite (what the driver runs), then the default bench line.
tag=${1:-full}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -q --capture=sys -m gpu --durations=8 > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest.log; tail -25 $out/pytest.log | cut -c1-220
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], (d["roofline"]["traffic_detail"] or {}).get("fetch_factor_reproducing_k_vec_FPlainQ"))
    print({k: round(v["frac"],3) for k,v in d["variant_rooflines"].items()})
    for k,e in d["extras"].items():
        print(k, e["rows"], e["nnz"], {m: (round(e[m]["ms_per_iteration"],4), round(e[m]["frac"],3)) for m in e if isinstance(e[m], dict) and "ms_per_iteration" in e[m]}, "spmv", round(e["spmv_back_to_back"]["ms"]*1e3,1), "us", round(e["spmv_back_to_back"]["frac"],3), e["plan"]["sell_rows"])
    print("cpu", d["cpu_baseline"], d["cpu_baseline_multicore"])
except Exception as e:
    print("bench parse failed", e); print(open("$out/bench.err").read()[-3000:])
PY
tail -12 $out/bench.err | cut -c1-200
