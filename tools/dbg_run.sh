#!/bin/bash
cd $GRAFT_REPO_ROOT
for sz in "--width 1280 --height 1024 --steps 60 --warmup 10 --seq-frames 16" "--steps 300 --warmup 30"; do
for tr in "" "--trials 1"; do
  LSDHIP_TRACK_DEBUG=1 python bench.py $sz --no-cpu-baseline --no-throughput-mode --no-pcie-leg $tr > /tmp/o.json 2> /tmp/e.txt
  grep TRACKDBG /tmp/e.txt | tail -1
  python -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('   [$tr] %s fps %.0f track_fps %.0f (%.1f us) map %.1f us evals %.2f launches %.2f avg_launch_us %.2f' % (d['config']['workload'][:9], d['value'], d['track_fps'], 1e6/d['track_fps'], d['ms_per_step']*1e3-1e6/d['track_fps'], d['lm_evaluations_per_frame'], d['track_launches_per_frame'], d['roofline']['avg_launch_us']))"
done; done
