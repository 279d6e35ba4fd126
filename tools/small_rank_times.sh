#!/bin/bash
# ms/iteration of a 200 k-row rank (1/8 of Transport) in the communication modes one GPU can run
export BICG_P2P_TIMEOUT_MS=3000 BENCH_WATCHDOG_S=200
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.1f us' % (1e3*d['value']), '|', d['config']['transport'][:50])"; }
B="python bench.py --rows 200264 --steps 400 --warmup 40 --no-cpu-baseline --no-variants --no-extras --no-traffic"
for m in bicgstab pipe_bicgstab ca_bicgstab; do
  timeout 100 $B --method $m 2>/dev/null | show "single        $m"
  timeout 100 $B --method $m --force-comm --transport auto 2>/dev/null | show "p2p fused     $m"
  timeout 100 $B --method $m --force-comm --transport rccl 2>/dev/null | show "rccl (1 rank) $m"
done
for m in bicgstab ca_bicgstab; do
  BICG_PERSIST=0 timeout 100 $B --method $m 2>/dev/null | show "single, multi-launch form            $m"
  BICG_PERSIST=0 timeout 100 $B --method $m --force-comm --transport auto 2>/dev/null | show "p2p, multi-launch form               $m"
done
# the multi-launch forms of the pipelined iteration (the default above is ONE persistent launch per chunk, bicg_persist.hip)
BICG_PERSIST=0 timeout 100 $B --method pipe_bicgstab 2>/dev/null | show "single, two launches per iteration   pipe_bicgstab"
BICG_PERSIST=0 timeout 100 $B --method pipe_bicgstab --force-comm --transport auto 2>/dev/null | show "p2p, two launches per iteration      pipe_bicgstab"
BICG_PERSIST=0 BICG_PLAN=fuse-pipe=0 timeout 100 $B --method pipe_bicgstab 2>/dev/null | show "single, phases as separate kernels  pipe_bicgstab"
BICG_PERSIST=0 BICG_PLAN=fuse-pipe=0 timeout 100 $B --method pipe_bicgstab --force-comm --transport auto 2>/dev/null | show "p2p, phases as separate kernels     pipe_bicgstab"
