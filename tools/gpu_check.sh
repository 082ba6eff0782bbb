#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench variants, rocprofv3 kernel trace.  Usage: tools/gpu_check.sh [tag]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
for CAP in 304 152; do
  LSDHIP_TRACK_CAP=$CAP timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_c${CAP}.json 2> $OUT/bench_c${CAP}.err
  echo "CAP=$CAP: $(python -c "import json,sys; d=json.load(open('$OUT/bench_c${CAP}.json')); print(d['value'], d['track_fps'], d['depth_mpix_per_s'], d['roofline'])" 2>&1 | tail -1)"
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline-events --no-throughput-mode > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -30
