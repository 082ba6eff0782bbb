#!/bin/bash
# same-box A/B of builds gpurun_in/liblsdhip_<name>.so ($@ = names) on the bench loop
cd $GRAFT_REPO_ROOT
cp lsd_slam_amd/liblsdhip.so /tmp/liblsdhip_keep.so
for rep in 1 2; do
for lib in "$@"; do
  cp gpurun_in/liblsdhip_$lib.so lsd_slam_amd/liblsdhip.so
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-throughput-mode --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib: fps %.0f track_fps %.0f ms %.4f depth_mpix %.0f observe_us %.2f ok %s' % (d['value'], d['track_fps'], d['ms_per_step'], d['depth_mpix_per_s'], d['roofline_depth']['avg_launch_us'], d['validation']['ok']))"
done; done
cp /tmp/liblsdhip_keep.so lsd_slam_amd/liblsdhip.so
