#!/bin/bash
# round 5, sixth GPU session: regulariser with compacted K6 centres (A/B against LSD_REG_NO_COMPACT), idepth pyramid with 32x32 tiles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s6
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log; grep -n "the reference's own b error" $O/pytest.log | cut -c1-260 | head -24
B=$R/lsd_slam_amd/liblsdhip_noregcompact.so
for rep in 1 2; do
  echo "--- 4K regulariser, compacted"; timeout 200 python tools/bench_bands.py --native --bands 1 2>&1 | tail -1 | cut -c1-330
  echo "--- 4K regulariser, round-4 form"; LD_PRELOAD=$B LSDHIP_LIB=$B timeout 200 python tools/bench_bands.py --native --bands 1 2>&1 | tail -1 | cut -c1-330
done | tee $O/bands.txt
ms() { python tools/bench_multiseq.py --S $2 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x.startswith('S')][0]; r=d[k].get('roofline') or {}; print(d['tag'], k, round(d[k]['frames_s']), round(d[k]['frames_s_block_until_mapped']), d[k]['replicas_bit_identical'], d[k]['tracked_good'], {kk: round(v['avg_launch_us']) for kk, v in r.items() if isinstance(v, dict)})"; }
for rep in 1 2; do
  ms compact 32
  LD_PRELOAD=$B LSDHIP_LIB=$B ms nocompact 32
done
for k in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs 2> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single', d['value'], d['ms_per_step'], d.get('keyframe_ms'), d['validation']['ok'])"
  LD_PRELOAD=$B LSDHIP_LIB=$B timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs 2> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single nocompact', d['value'], d['ms_per_step'], d.get('keyframe_ms'), d['validation']['ok'])"
done
