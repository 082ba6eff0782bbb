#!/bin/bash
# same-box A/B of two builds of liblsdhip.so (gpurun_in/liblsdhip_old.so / _new.so) on the bench loop
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in old new; do
for cfg in "1 0 0,0,0,0,0" "5 104 0,4,5,6,0"; do
  set -- $cfg
  cp $GRAFT_REPO_ROOT/gpurun_in/liblsdhip_$lib.so $GRAFT_REPO_ROOT/lsd_slam_amd/liblsdhip.so; LSDHIP_SPEC_LEVELS=$3 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-throughput-mode --no-pcie-leg --trials $1 --trial-cap $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib trials $1: fps %.0f track_fps %.0f ms %.4f launches %.2f avg_launch_us %.2f ok %s' % (d['value'], d['track_fps'], d['ms_per_step'], d['track_launches_per_frame'], d['roofline']['avg_launch_us'], d['validation']['ok']))"
done; done; done
