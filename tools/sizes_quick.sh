#!/bin/bash
# quick look at the other frame sizes (no CPU baseline); $1 = extra bench flags
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s: fps %.0f track_fps %.0f evals %.2f launches %.2f avg_launch_us %.2f depth_mpix %.0f ok %s' % (d['config']['workload'][:9], d['value'], d['track_fps'], d['lm_evaluations_per_frame'], d['track_launches_per_frame'], d['roofline']['avg_launch_us'], d['depth_mpix_per_s'], d['validation']['ok']))"; }
for extra in "" "--trials 1"; do
echo "== flags: $extra"
run --steps 300 --warmup 30 $extra
run --width 1280 --height 1024 --steps 60 --warmup 10 --seq-frames 16 $extra
run --width 1280 --height 1024 --scene S2 --steps 60 --warmup 10 --seq-frames 16 $extra
run --width 3840 --height 2160 --steps 24 --warmup 6 --seq-frames 6 $extra
done
