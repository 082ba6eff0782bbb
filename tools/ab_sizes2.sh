#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "A" "A --trials 1" "B" "B --trials 1"; do
  set -- $cfg
  if [ $1 = A ]; then sz="--steps 300 --warmup 30"; else sz="--width 1280 --height 1024 --steps 240 --warmup 20 --seq-frames 16"; fi
  python bench.py $sz --no-cpu-baseline --no-throughput-mode --no-pcie-leg $2 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$cfg] %s fps %.0f track_fps %.0f evals %.2f launches %.2f avg_launch_us %.2f ok %s' % (d['config']['workload'][:9], d['value'], d['track_fps'], d['lm_evaluations_per_frame'], d['track_launches_per_frame'], d['roofline']['avg_launch_us'], d['validation']['ok']))"
done; done
