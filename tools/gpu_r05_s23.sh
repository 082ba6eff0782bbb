#!/bin/bash
# round 5, twenty-third GPU session: fused multiply-adds in the tolerance-only K2 / K3 formulas of the residual evaluation — parity tests, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s23
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error" $O/pytest.log | tail -5
H=$R/lsd_slam_amd/liblsdhip_head.so
for rep in 1 2; do
echo "--- evaluation launches (64 jobs): head / new"
LD_PRELOAD=$H LSDHIP_LIB=$H timeout 200 python tools/bench_eval.py 2>/dev/null | cut -c1-200
timeout 200 python tools/bench_eval.py 2>/dev/null | cut -c1-200
done
ms() { python tools/bench_multiseq.py --S $2 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x.startswith('S')][0]; r=d[k].get('roofline') or {}; print(d['tag'], k, round(d[k]['frames_s']), round(d[k]['frames_s_block_until_mapped']), d[k]['replicas_bit_identical'], d[k]['tracked_good'], {kk: round(v['avg_launch_us']) for kk, v in r.items() if isinstance(v, dict)})"; }
for rep in 1 2 3; do
  LD_PRELOAD=$H LSDHIP_LIB=$H ms head 32
  ms new 32
done
for v in head new; do
  if [ $v = head ]; then export LD_PRELOAD=$H LSDHIP_LIB=$H; else unset LD_PRELOAD LSDHIP_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --no-extra-configs 2>> $O/bench.err | tee $O/bench_$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single', '$v', d['value'], d['ms_per_step'], d['validation'], (d.get('roofline_throughput_mode') or {}).get('frac'), ((d.get('roofline_throughput_mode') or {}).get('level1_evaluation') or {}).get('frac'))"
done
unset LD_PRELOAD LSDHIP_LIB
