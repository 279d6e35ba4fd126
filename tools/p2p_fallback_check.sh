export BICG_P2P_TIMEOUT_MS=2000 BENCH_WATCHDOG_S=200
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', d['value'], '| transport:', c['transport'][:110], '| genuine', c['iterations_genuine'], c['relres_after_timed_region'], c['true_relres_after_timed_region'])"; }
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants 2>/dev/null | show n1
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517"
timeout 250 $TR bench.py --gpus 2 --rows 400528 --transport host-p2p --steps 100 --warmup 10 --no-cpu-baseline --no-variants 2>gpurun_out/f0.err | show p2p_ok
BICG_P2P_FAULT_AFTER=150 timeout 250 $TR bench.py --gpus 2 --rows 400528 --transport host-p2p --steps 100 --warmup 10 --no-cpu-baseline --no-variants 2>gpurun_out/f1.err | show p2p_fault
grep "bench \|bicgstab_hip" gpurun_out/f1.err | tail
BICG_P2P_TIMEOUT_MS=0.0001 timeout 250 $TR bench.py --gpus 2 --rows 400528 --transport host-p2p --steps 100 --warmup 10 --no-cpu-baseline --no-variants 2>gpurun_out/f2.err | show p2p_selftest_fail
grep "bench \|bicgstab_hip" gpurun_out/f2.err | tail
