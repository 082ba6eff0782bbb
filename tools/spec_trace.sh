#!/bin/bash
# per-launch durations of k_track_step, in launch order, for a few frames of the bench loop (rocprofv3 kernel trace)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/spec_trace
mkdir -p $OUT
cd /tmp
for cfg in "1 0 0,0,0,0,0" "5 104 0,4,5,6,0"; do
  set -- $cfg
  rm -rf /tmp/st_$1
  LSDHIP_SPEC_LEVELS=$3 rocprofv3 --kernel-trace --output-format csv -d /tmp/st_$1 -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-roofline-events --trials $1 --trial-cap $2 > /dev/null 2>&1
  python - $1 > $OUT/trials_$1.txt <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/st_%s/**/*kernel_trace.csv' % sys.argv[1], recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# frames = runs of k_track_step between other kernels
seqs, cur = [], []
prev_end = None
for r in rows:
    n = r['Kernel_Name']
    if 'k_track_step' in n:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        cur.append(((e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0, int(r.get('Grid_Size_X', r.get('Grid_Size', 0)))))
        prev_end = e
    else:
        if cur: seqs.append(cur); cur = []
        prev_end = int(r['End_Timestamp'])
for q in seqs[40:46]:
    print(' '.join('%.1f(+%.1f)' % (d, g) for d, g, _ in q))
    print('   sum %.1f us, %d launches, grid %s' % (sum(d + g for d, g, _ in q), len(q), q[0][2]))
PY
  cat $OUT/trials_$1.txt
done
