#!/bin/bash
timeout 600 python -m pytest tests/test_full_size.py tests/test_gpu_parity.py -q --capture=sys -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|ERROR:|^E  " | tail -5 | cut -c1-200
BICG_SELL_WINDOW=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -q --capture=sys -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|ERROR:|^E  " | tail -4 | cut -c1-200
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "ms/iter", round(d["value"],4), "spmv us", round(1e3*d["roofline"]["avg_launch_ms"],1), "b2b", round(1e3*(d["roofline"]["back_to_back_spmv_ms"] or 0),1))'
for srt in 0 1; do
  BICG_SELL_SORT=$srt timeout 200 $B --workload fem_like 2>/dev/null | python -c "$show" "fem_like sort=$srt"
  BICG_SELL_SORT=$srt timeout 200 $B --workload fem_like --method pipe_bicgstab 2>/dev/null | python -c "$show" "fem_like pipe sort=$srt"
  BICG_SELL_SORT=$srt BICG_FUSE_PIPE=1 timeout 200 $B --workload fem_like --method pipe_bicgstab 2>/dev/null | python -c "$show" "fem_like pipe fused sort=$srt"
done
