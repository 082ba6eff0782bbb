#!/bin/bash
# round 5, second GPU session: suite; evaluation launch (occupancy 4) against the round-4 form; shared keyframe change; walk kernel at 128 registers
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_ab2
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for i in 1 2; do timeout 300 python -m pytest tests/test_dataset_gpu.py -q -k deterministic >> $O/pytest_det.log 2>&1; echo "determinism run $i rc=$?"; done
tail -4 $O/pytest_det.log
B=$R/lsd_slam_amd/liblsdhip_r04eval.so
W=$R/lsd_slam_amd/liblsdhip_walk4.so
echo "--- eval launch: new (1024 strips, occupancy 4)"; timeout 200 python tools/bench_eval.py --levels 3,2,1 2> $O/eval_new.err | tee $O/eval_new.json
echo "--- eval launch: new lib, 768 strips"; LSDHIP_BATCH_WGS=768 timeout 200 python tools/bench_eval.py --levels 1 2>> $O/eval_new.err | tee -a $O/eval_new.json
echo "--- eval launch: round-4 form (768 strips)"; LD_PRELOAD=$B LSDHIP_LIB=$B LSDHIP_BATCH_WGS=768 timeout 200 python tools/bench_eval.py --levels 3,2,1 2> $O/eval_r04.err | tee $O/eval_r04.json
echo "--- batches"; timeout 200 python tools/bench_batch.py --batches 32,64 --rounds 10 2> $O/batch_new.err | tee $O/batch_new.json
LD_PRELOAD=$B LSDHIP_LIB=$B LSDHIP_BATCH_WGS=768 timeout 200 python tools/bench_batch.py --batches 32,64 --rounds 10 2> $O/batch_r04.err | tee $O/batch_r04.json
ms() { python tools/bench_multiseq.py --S 32 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['tag'], round(d['S32']['frames_s']), round(d['S32']['frames_s_block_until_mapped']), d['S32']['replicas_bit_identical'], d['S32']['tracked_good'])"; }
for rep in 1 2; do
  ms new
  LSDHIP_KF_SHARED=0 ms kf_per_sequence
  LD_PRELOAD=$W LSDHIP_LIB=$W ms walk_occ4
  LD_PRELOAD=$B LSDHIP_LIB=$B LSDHIP_BATCH_WGS=768 ms eval_r04
done
for k in 1 0 1 0; do
  LSDHIP_KF_SHARED=$k timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs 2> $O/bench_kf$k.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('KF_SHARED=$k', d['value'], d['ms_per_step'], d.get('keyframe_ms'), d['validation']['ok'])"
done
