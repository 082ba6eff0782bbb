#!/bin/bash
# round 5: run-to-run determinism of the C++ loop, per frame and per chunk of 10 frames, one-stream and pipelined
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05_det
for rep in 1 2; do
timeout 300 python tools/determinism_frames.py --reps 10 2>&1 | tail -6 | cut -c1-300 | tee -a gpurun_out/r05_det/frames.txt
timeout 300 python tools/determinism_frames.py --reps 10 --pipelined 2>&1 | tail -6 | cut -c1-300 | tee -a gpurun_out/r05_det/frames.txt
timeout 300 python tools/determinism_chunks.py --reps 10 2>&1 | tail -4 | tee -a gpurun_out/r05_det/chunks.txt
timeout 300 python tools/determinism_chunks.py --reps 10 --pipelined 2>&1 | tail -4 | tee -a gpurun_out/r05_det/chunks.txt
done
