#!/bin/bash
# round 5: last check of the committed tree — build entry, smoke, the driver's bench invocation, suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_last
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$O/bench.json'))
print(d['metric'][:60], round(d['value']), d['unit'], d['ms_per_step'], 'kf_ms', d.get('keyframe_ms'), 'valid', d['validation']['ok'])
print('roofline', {k: d['roofline'][k] for k in ('bound','achieved','peak','frac','traffic','traffic_source')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('pipelined',{}).get('value'), d['speedup_vs_cpu_baseline']['pipelined'], d['speedup_vs_cpu_baseline']['block_until_mapped'])
ms=d['extra_configs']['multi_seq']; print('S8', round(ms['S8']['frames_s']), 'S32', round(ms['S32']['frames_s']))
print('tm', d['roofline_throughput_mode']['frac'], d['roofline_throughput_mode']['level1_evaluation']['frac'])
"
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed" $O/pytest.log | tail -2
