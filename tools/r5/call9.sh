#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5c9
timeout 150 python tools/r5/hang_probe2.py > gpurun_out/r5c9/probe2.txt 2>&1
tail -6 gpurun_out/r5c9/probe2.txt
timeout 400 python -m pytest tests/test_switching.py tests/test_dropin_host.py tests/test_full_size.py -x -q -m gpu 2>&1 | tail -12
