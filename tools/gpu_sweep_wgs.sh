#!/bin/bash
# developer sweep: strips x jobs of a throughput-mode evaluation launch (LSDHIP_BATCH_WGS) against whole-batch time and the S-sequence loop
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in 768 1152 1536 2304 768; do
  echo "LSDHIP_BATCH_WGS=$v"
  LSDHIP_BATCH_WGS=$v python bench.py --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline_throughput_mode']; print(' whole batches frac %.4f stream_ms %.3f L1 eval %.1f us' % (r['frac'], r['stream_ms'], r['evaluation_launch_by_level']['L1']['us_per_launch']))
m=d['extra_configs'].get('multi_seq'); print(' multi', {k:(round(v['frames_s']),v['tracked_good']) for k,v in m.items() if k.startswith('S')} if 'error' not in m else m)"
done
