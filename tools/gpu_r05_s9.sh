#!/bin/bash
# round 5, ninth GPU session: exact reciprocal (all 2^32 inputs), branch-free regulariser loop — same-box A/B of three builds, suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s9
mkdir -p $O
timeout 200 lsd_slam_amd/rcp_exhaustive.bin | tee $O/rcp_exhaustive.json
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
ms() { python tools/bench_multiseq.py --S $2 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x.startswith('S')][0]; r=d[k].get('roofline') or {}; print(d['tag'], k, round(d[k]['frames_s']), round(d[k]['frames_s_block_until_mapped']), d[k]['replicas_bit_identical'], d[k]['tracked_good'], {kk: round(v['avg_launch_us']) for kk, v in r.items() if isinstance(v, dict)})"; }
H=$R/lsd_slam_amd/liblsdhip_head.so
I=$R/lsd_slam_amd/liblsdhip_ieee.so
for rep in 1 2; do
  echo "--- 4K regulariser pass: head / ieee (branch-free, IEEE division) / new"
  LD_PRELOAD=$H LSDHIP_LIB=$H timeout 200 python tools/bench_bands.py --native --bands 1 2>&1 | tail -1 | cut -c1-200
  LD_PRELOAD=$I LSDHIP_LIB=$I timeout 200 python tools/bench_bands.py --native --bands 1 2>&1 | tail -1 | cut -c1-200
  timeout 200 python tools/bench_bands.py --native --bands 1 2>&1 | tail -1 | cut -c1-200
  LD_PRELOAD=$H LSDHIP_LIB=$H ms head 32
  LD_PRELOAD=$I LSDHIP_LIB=$I ms ieee 32
  ms new 32
done
for v in head new; do
  if [ $v = head ]; then export LD_PRELOAD=$H LSDHIP_LIB=$H; else unset LD_PRELOAD LSDHIP_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --no-extra-configs 2>> $O/bench.err | tee $O/bench_$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single', '$v', d['value'], d['ms_per_step'], d.get('keyframe_ms'), d['validation']['ok'], (d.get('roofline_throughput_mode') or {}).get('frac'), ((d.get('roofline_throughput_mode') or {}).get('level1_evaluation') or {}).get('frac'))"
done
unset LD_PRELOAD LSDHIP_LIB
