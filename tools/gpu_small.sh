#!/bin/bash
out=gpurun_out/${1:-small}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_multirank.py tests/test_comm_path_one_gpu.py tests/test_shifted.py -q --capture=sys -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $out/pytest.log | tail -8 | cut -c1-200
tools/small_rank_times.sh 2>&1 | tee $out/small_rank_times.txt
cd /tmp && export TMPDIR=/tmp
for mode in single p2p; do
  extra=""; [ $mode = p2p ] && extra="--force-comm --transport auto"
  timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$out/prof_$mode -o run --output-format csv -- python /root/repo/bench.py --rows 200264 --steps 400 --warmup 40 --no-cpu-baseline --no-variants --no-extras --no-traffic --method pipe_bicgstab $extra > /root/repo/$out/prof_$mode.log 2>&1
  f=$(find /root/repo/$out/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f /root/repo/$out/kernel_stats_small_$mode.csv && head -6 /root/repo/$out/kernel_stats_small_$mode.csv | cut -c1-170
  rm -rf /root/repo/$out/prof_$mode
done
