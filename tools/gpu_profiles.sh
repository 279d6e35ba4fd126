#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries for profiles/rNN: headline (plain), pipelined (non-temporal SpMV
# instantiations), FEM-like (jagged slices + x windows) and the same matrix on the CSR row-block kernel
tag=${1:-prof}; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic"
run() {   # name, env assignment (or ''), bench args...
  name=$1; envs=$2; shift 2
  rm -rf /tmp/prof_$name
  env $envs timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run --output-format csv -- $B "$@" > $out/prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $out/kernel_stats_$name.csv; echo "== $name"; head -6 $out/kernel_stats_$name.csv | cut -c1-170; else echo "== $name: no stats"; tail -5 $out/prof_$name.log; fi
}
want() { [ -z "$ONLY" ] || echo " $ONLY " | grep -q " $1 "; }
want plain && run plain "BICG_X=1"
want pipe_nt && run pipe_nt "BICG_X=1" --method pipe_bicgstab
want pipe_separate && run pipe_separate "BICG_FUSE_PIPE=0" --method pipe_bicgstab
want ca_nt && run ca_nt "BICG_X=1" --method ca_bicgstab
want fem_like_window && run fem_like_window "BICG_X=1" --workload fem_like
want fem_like_pipe && run fem_like_pipe "BICG_X=1" --workload fem_like --method pipe_bicgstab
want fem_like_csr && run fem_like_csr "BICG_NO_SELL=1" --workload fem_like
true
