#!/bin/bash
out=gpurun_out/${1:-pipe}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_multirank.py tests/test_comm_path_one_gpu.py tests/test_shifted.py tests/test_switching.py tests/test_dropin_cache.py -q --capture=sys -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $out/pytest.log | tail -12 | cut -c1-200
BICG_FUSE_PIPE=0 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k pipe 2>&1 | tail -2
tools/small_rank_times.sh 2>&1 | tee $out/small_rank_times.txt
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic"
for m in pipe_bicgstab pipe_bicgstab_rr; do
  for f in 1 0; do BICG_FUSE_PIPE=$f timeout 200 $B --method $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full size $m fuse=$f', round(d['value'],4), d['config']['iterations_genuine'])"; done
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$out/prof -o run --output-format csv -- python /root/repo/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants --no-extras --no-traffic --method pipe_bicgstab > /root/repo/$out/prof.log 2>&1
cd /root/repo; f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_pipe.csv && head -8 $out/kernel_stats_pipe.csv | cut -c1-180; rm -rf $out/prof
