#!/bin/bash
# round 5, eighth GPU session: exhaustive reciprocal check, tracking stream priority, suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s8
mkdir -p $O
timeout 120 lsd_slam_amd/rcp_exhaustive.bin | tee $O/rcp_exhaustive.json
ms() { python tools/bench_multiseq.py --S $2 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x.startswith('S')][0]; r=d[k].get('roofline') or {}; print(d['tag'], k, round(d[k]['frames_s']), round(d[k]['frames_s_block_until_mapped']), d[k]['replicas_bit_identical'], d[k]['tracked_good'], {kk: round(v['avg_launch_us']) for kk, v in r.items() if isinstance(v, dict)})"; }
for rep in 1 2; do
  ms base 32
  LSDHIP_TRACK_PRIO=1 ms trackprio 32
done
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
