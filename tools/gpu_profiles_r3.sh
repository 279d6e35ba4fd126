#!/bin/bash
# Round-3 evidence in one GPU call: driver-flag bench line, long bench line, rocprofv3 kernel stats of the headline, of the
# persistent small-rank iteration and of the long-row SpMV, STREAM numbers. tools/collect_r3.py copies the summaries to profiles/r03/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
if [ -z "$PROFILES_ONLY" ]; then
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1_driver_flags.json 2> $OUT/bench_n1_driver_flags.err
python bench.py --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err
fi
BICG_STREAM_VERBOSE=1 python -c "
import sys; sys.path.insert(0, '.')
from mpi_bicgstab_amd import hipsolver as H
H.lib().bicg_comm_init_single(0)
for k in ('copy', 'triad', 'read8', 'read16'):
    print(k, '%.1f GB/s' % max(H.stream_bench(k, 1 << 30, 20) for _ in range(2)))
" > $OUT/stream.txt 2>&1
cd /tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o $name --output-format csv -- "$@" > $OUT/prof_$name.log 2>&1
  cp $(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/prof_$name
}
B="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants --no-extras --no-traffic --no-stream"
prof headline_plain $B
prof headline_pipe $B --method pipe_bicgstab
prof headline_ca $B --method ca_bicgstab
prof small_rank_pipe_persist $B --rows 200264 --method pipe_bicgstab --steps 400
prof small_rank_plain_persist $B --rows 200264 --method bicgstab --steps 400
prof banded_b512_plain $B --workload banded --half-bandwidth 512
prof shifted16 python $R/tools/ab_shift.py
ls -la $OUT
