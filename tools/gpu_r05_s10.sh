#!/bin/bash
# round 5, tenth GPU session: Sim3 LM on the device (first run), guarded exact reciprocal, one-read regulariser entries — suite + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s10
mkdir -p $O
timeout 200 lsd_slam_amd/rcp_exhaustive.bin | tee $O/rcp_exhaustive.json | cut -c1-400
timeout 300 python -m pytest tests/test_sim3_gpu.py -m gpu -q -x > $O/pytest_sim3.log 2>&1; echo "pytest sim3 rc=$?"; tail -15 $O/pytest_sim3.log
H=$R/lsd_slam_amd/liblsdhip_head.so
I=$R/lsd_slam_amd/liblsdhip_ieee.so
echo "--- sim3 timing: head / new"
LD_PRELOAD=$H LSDHIP_LIB=$H timeout 200 python tools/bench_sim3.py 2>&1 | tail -2 | tee $O/sim3_head.json
timeout 200 python tools/bench_sim3.py 2>&1 | tail -2 | tee $O/sim3_new.json
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_sim3_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
ms() { python tools/bench_multiseq.py --S $2 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x.startswith('S')][0]; r=d[k].get('roofline') or {}; print(d['tag'], k, round(d[k]['frames_s']), round(d[k]['frames_s_block_until_mapped']), d[k]['replicas_bit_identical'], d[k]['tracked_good'], {kk: round(v['avg_launch_us']) for kk, v in r.items() if isinstance(v, dict)})"; }
for rep in 1 2; do
  LD_PRELOAD=$H LSDHIP_LIB=$H ms head 32
  LD_PRELOAD=$I LSDHIP_LIB=$I ms ieee 32
  ms new 32
done
for v in head new; do
  if [ $v = head ]; then export LD_PRELOAD=$H LSDHIP_LIB=$H; else unset LD_PRELOAD LSDHIP_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg 2>> $O/bench.err | tee $O/bench_$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d.get('extra_configs') or {}; print('single', '$v', d['value'], d['ms_per_step'], d.get('keyframe_ms'), d['validation']['ok'], (d.get('roofline_throughput_mode') or {}).get('frac'), ((d.get('roofline_throughput_mode') or {}).get('level1_evaluation') or {}).get('frac'), d.get('roofline_depth'), {k: (v.get('ms_per_pass') if isinstance(v, dict) else None) for k, v in e.items()})"
done
unset LD_PRELOAD LSDHIP_LIB
