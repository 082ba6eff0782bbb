#!/bin/bash
# A/B of the reject-chain speculation on the bench loop: "trials cap levels(l0,l1,l2,l3,l4)"
cd $GRAFT_REPO_ROOT
while read -r tr cap lv caps; do
  [ -z "$tr" ] && continue
  LSDHIP_SPEC_LEVELS=$lv LSDHIP_SPEC_CAPS=${caps:-0,0,0,0,0} python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-throughput-mode --no-pcie-leg --trials $tr --trial-cap $cap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('trials $tr cap $cap levels $lv caps $caps: fps %.0f track_fps %.0f ms %.4f evals %.2f launches %.2f avg_launch_us %.2f ok %s' % (d['value'], d['track_fps'], d['ms_per_step'], d['lm_evaluations_per_frame'], d['track_launches_per_frame'], d['roofline']['avg_launch_us'], d['validation']['ok']))"
done < ${SPEC_CFG_FILE:-tools/spec_cfgs.txt}
