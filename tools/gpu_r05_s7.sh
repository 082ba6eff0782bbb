#!/bin/bash
# round 5, seventh GPU session: observe select with four pixels per lane; compaction of the regulariser's centres at 3840x2160 (30 % valid)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s7
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
B=$R/lsd_slam_amd/liblsdhip_noregcompact.so
python tools/bench_bands.py --native --bands 1 > /dev/null 2>&1   # warm the box
for rep in 1 2 3; do
  echo -n "4K compact   "; timeout 200 python tools/bench_bands.py --native --bands 1 --passes 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_pass'], d['algorithmic_GBps'])"
  echo -n "4K nocompact "; LD_PRELOAD=$B LSDHIP_LIB=$B timeout 200 python tools/bench_bands.py --native --bands 1 --passes 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_pass'], d['algorithmic_GBps'])"
done | tee $O/bands.txt
ms() { python tools/bench_multiseq.py --S $2 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x.startswith('S')][0]; r=d[k].get('roofline') or {}; print(d['tag'], k, round(d[k]['frames_s']), round(d[k]['frames_s_block_until_mapped']), d[k]['replicas_bit_identical'], d[k]['tracked_good'], {kk: round(v['avg_launch_us']) for kk, v in r.items() if isinstance(v, dict)})"; }
for rep in 1 2; do
  ms select4 32
  LD_PRELOAD=$B LSDHIP_LIB=$B ms select4_nocompact 32
done
ms select4 8
