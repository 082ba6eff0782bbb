#!/bin/bash
# round 5, twenty-fourth / twenty-fifth GPU session: regulariser neighbour loop with its constants in scalar registers, counter as a float — depth tests, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s25
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multiseq_gpu.py tests/test_bands_gpu.py tests/test_rcp_gpu.py tests/test_sequence_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error" $O/pytest.log | tail -3
H=$R/lsd_slam_amd/liblsdhip_head.so
for rep in 1 2; do
for v in head new; do
  if [ $v = head ]; then export LD_PRELOAD=$H LSDHIP_LIB=$H; else unset LD_PRELOAD LSDHIP_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --no-throughput-mode 2>> $O/bench.err | tee $O/bench_${v}_$rep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['extra_configs']; r=e['reg_3840x2160']; ms=e['multi_seq']; print('$v', round(d['value']), 'kf_ms', round(d['keyframe_ms'],4), 'reg4k', round(r['full_frame']['ms_per_pass'],4), round(r['full_frame']['frac'],4), 'bands', round(r['bands_vs_full_frame'],3), 'S8', round(ms['S8']['frames_s']), 'S32', round(ms['S32']['frames_s']), {k: round(v['avg_launch_us']) for k, v in ms['S32']['roofline'].items() if isinstance(v, dict)})"
done
done
unset LD_PRELOAD LSDHIP_LIB
