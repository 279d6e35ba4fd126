#!/bin/bash
# pipelined iteration as 2 launches (phases in the SpMV epilogues) vs 4, by rank size and matrix
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "ms/iter", round(d["value"],4), d["config"].get("iterations_genuine"))'
for rows in ${ROWS:-400000 800000 1602111}; do
  for f in 0 1; do BICG_FUSE_PIPE=$f timeout 200 $B --method pipe_bicgstab --rows $rows 2>/dev/null | python -c "$show" "transport rows=$rows fuse=$f"; done
done
for f in 0 1; do BICG_FUSE_PIPE=$f timeout 300 $B --method pipe_bicgstab --workload laplace7 --grid 256 --steps 60 2>/dev/null | python -c "$show" "laplace256 fuse=$f"; done
for f in 0 1; do BICG_FUSE_PIPE=$f timeout 200 $B --method pipe_bicgstab --workload banded --half-bandwidth 8 2>/dev/null | python -c "$show" "banded8 fuse=$f"; done
for f in 0 1; do BICG_FUSE_PIPE=$f timeout 200 $B --method pipe_bicgstab --workload fem_like 2>/dev/null | python -c "$show" "fem_like fuse=$f"; done
