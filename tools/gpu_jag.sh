#!/bin/bash
# jagged slices / x windows: parity, then time per workload and layout choice
out=gpurun_out/${1:-jag}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_shifted.py tests/test_multirank.py -q --capture=sys -m gpu -x > $out/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|ERROR:|^E  " $out/pytest.log | tail -8 | cut -c1-200
BICG_SELL_WINDOW=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -q --capture=sys -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|ERROR:" | tail -4 | cut -c1-200
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "ms/iter", round(d["value"],4), "spmv us", round(1e3*d["roofline"]["avg_launch_ms"],1), "b2b", round(1e3*(d["roofline"]["back_to_back_spmv_ms"] or 0),1), "TB/s", round(d["roofline"]["achieved"]), d["config"].get("iterations_genuine"))'
for w in ${WINS:-auto 0 1}; do
  if [ $w = auto ]; then unset BICG_SELL_WINDOW; else export BICG_SELL_WINDOW=$w; fi
  timeout 200 $B --workload fem_like 2>/dev/null | python -c "$show" "fem_like win=$w"
  timeout 200 $B 2>/dev/null | python -c "$show" "transport win=$w"
  timeout 200 $B --method pipe_bicgstab 2>/dev/null | python -c "$show" "transport pipe win=$w"
  timeout 200 $B --workload laplace7 --grid 256 2>/dev/null | python -c "$show" "laplace256 win=$w"
  timeout 200 $B --workload banded --half-bandwidth 64 2>/dev/null | python -c "$show" "banded64 win=$w"
done
