#!/bin/bash
# same-box A/B of environment switches on the bench loop: each line of $1 is "VAR=val VAR=val ..." (or "-" for none)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
while read -r envs; do
  [ -z "$envs" ] && continue
  [ "$envs" = "-" ] && e="" || e="$envs"
  env $e python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-throughput-mode --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$envs] fps %.0f track_fps %.0f ms %.4f launches %.2f avg_launch_us %.2f depth_mpix %.0f ok %s' % (d['value'], d['track_fps'], d['ms_per_step'], d['track_launches_per_frame'], d['roofline']['avg_launch_us'], d['depth_mpix_per_s'], d['validation']['ok']))"
done < $1
done
