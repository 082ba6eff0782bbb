#!/bin/bash
# round 5, seventeenth GPU session: Sim3 LM on the device, transcendental values in parallel lanes — tests, timing against the host-driven loop, phase stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s17
mkdir -p $O
timeout 300 python -m pytest tests/test_sim3_gpu.py -m gpu -q -x > $O/pytest_sim3.log 2>&1; echo "pytest sim3 rc=$?"; tail -15 $O/pytest_sim3.log
H=$R/lsd_slam_amd/liblsdhip_head.so
D=$R/lsd_slam_amd/liblsdhip_devtools.so
for rep in 1 2; do
echo "--- sim3 timing: head / new"
LD_PRELOAD=$H LSDHIP_LIB=$H timeout 200 python tools/bench_sim3.py 2>&1 | tail -2 | tee -a $O/sim3_head.json
timeout 200 python tools/bench_sim3.py 2>&1 | tail -2 | tee -a $O/sim3_new.json
done
LD_PRELOAD=$D LSDHIP_LIB=$D LSDHIP_S3_TRACE=1 timeout 200 python tools/bench_sim3.py > $O/sim3_trace.out 2> $O/sim3_trace.err
awk '{print $2, $3, $4, $5, $6}' $O/sim3_trace.err | sort | uniq -c | sort -rn | head -4
grep s3trace $O/sim3_trace.err | awk 'NR%20==1' | head -12
