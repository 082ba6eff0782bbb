#!/bin/bash
# round 5: the two-process IPC band test, repeated (after the null-stream memset fix)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05_bands
for rep in 1 2 3 4 5 6 7 8; do
timeout 400 python -m pytest tests/test_bands_gpu.py -m gpu -q -x 2>&1 | grep -a "passed\|failed" | tail -1 | tee -a gpurun_out/r05_bands/log.txt
done
