#!/bin/bash
# streaming policy of the matrix loads, re-measured: BICG_SELL_NT=0/1 on the full-size Transport-shaped problem
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic"
for m in bicgstab ca_bicgstab pipe_bicgstab pipe_bicgstab_rr; do
  for f in 0 1; do BICG_SELL_NT=$f timeout 200 $B --method $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m nt=$f', round(d['value'],4), round(d['roofline']['achieved']), d['config']['iterations_genuine'])"; done
done
