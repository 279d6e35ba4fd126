#!/bin/bash
# One gpurun call: GPU tests, the self-help path of the consumer-side finish, bench + kernel stats.
#   tools/gpu_check.sh <tag> [quick]
tag=${1:-chk}; out=gpurun_out/$tag; mkdir -p $out
export BENCH_WATCHDOG_S=300
if [ "$2" != "nobuildtests" ]; then
timeout 900 python -m pytest tests -q --capture=sys -m gpu --deselect tests/test_multirank_fullsize.py --deselect tests/test_bench_torchrun.py > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest.log; tail -15 $out/pytest.log
BICG_SPIN_TICKS=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -x -q -m gpu > $out/pytest_spin0.log 2>&1
echo "spin0 rc=$?" | tee -a $out/pytest_spin0.log; tail -5 $out/pytest_spin0.log
fi
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
timeout 300 $B > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["variants_ms_per_iteration"])
except Exception as e:
    print("bench parse failed", e); print(open("$out/bench.err").read()[-3000:])
PY
for g in 1024 1536; do
  BICG_VEC_GRID=$g timeout 200 $B --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vec_grid $g', d['value'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$out/prof -o run --output-format csv -- python /root/repo/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants > /root/repo/$out/prof.log 2>&1
cd /root/repo
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && head -12 $out/kernel_stats.csv | cut -c1-200
rm -rf $out/prof
