#!/bin/bash
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic"
for m in pipe_bicgstab; do
  for f in 1 0; do BICG_FUSE_PIPE=$f timeout 200 $B --method $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full size $m fuse=$f', round(d['value'],4), d['config']['iterations_genuine'])"; done
done
for r in 400000 800000; do for f in 1 0; do BICG_FUSE_PIPE=$f timeout 200 $B --method pipe_bicgstab --rows $r 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows $r fuse=$f', round(1e3*d['value'],1), 'us')"; done; done
