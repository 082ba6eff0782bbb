#!/bin/bash
# quick GPU iteration: parity tests + phase trace + one bench line.  Usage: tools/gpu_quick.sh <tag>
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
LSDHIP_LIB=lsd_slam_amd/liblsdhip_trace.so LSDHIP_TRACE_FILE=$OUT/trace.txt timeout 200 python tools/phase_trace.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print('fps', d['value'], 'track', d['track_fps'], 'depth', d['depth_mpix_per_s'], 'evals', d['lm_evaluations_per_frame'], 'launch_us', d['roofline'] and d['roofline']['avg_launch_us'])"
if [ -n "$PROF" ]; then
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline-events --no-throughput-mode > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof.err
  cd $GRAFT_REPO_ROOT
  python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-60s n=%5s avg=%8.1f us tot=%8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
  rm -rf $OUT/prof
fi
