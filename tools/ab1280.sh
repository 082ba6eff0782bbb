#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "0,0,0,0,0 " "0,1,1,5,0 " "0,1,4,5,0 " "0,0,0,0,0 --trials 1"; do
  set -- $cfg
  LSDHIP_SPEC_LEVELS=$1 python bench.py --width 1280 --height 1024 --steps 240 --warmup 20 --seq-frames 16 --no-cpu-baseline --no-throughput-mode --no-pcie-leg $2 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$cfg] fps %.0f track_fps %.0f evals %.2f launches %.2f avg_launch_us %.2f' % (d['value'], d['track_fps'], d['lm_evaluations_per_frame'], d['track_launches_per_frame'], d['roofline']['avg_launch_us']))"
done; done
