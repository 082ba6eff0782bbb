#!/bin/bash
# kernel timeline of two steady-state frames of the bench loop (rocprofv3 --kernel-trace): tools/gpu_timeline.sh <tag> [bench flags]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -- python $R/bench.py --steps 60 --warmup 10 --repeats 1 --no-cpu-baseline --no-roofline-events --no-throughput-mode --no-pcie-leg --no-extra-configs "$@" > $R/gpurun_out/tl_$TAG.json 2> $R/gpurun_out/tl_$TAG.err
cd $R
python tools/timeline.py gpurun_out/tl_$TAG --all > gpurun_out/timeline_$TAG.txt 2>&1
rm -rf gpurun_out/tl_$TAG
tail -60 gpurun_out/timeline_$TAG.txt
