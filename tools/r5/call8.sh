#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5c8
timeout 200 python -m pytest tests/test_switching.py -x -q -m gpu --timeout=50 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r5c8/switching.txt
cat gpurun_out/r5c8/switching.txt
