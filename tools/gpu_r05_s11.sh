#!/bin/bash
# round 5, eleventh GPU session: where a fused Sim3 launch spends its time (devtools build, phase stamps)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s11
mkdir -p $O
D=$R/lsd_slam_amd/liblsdhip_devtools.so
LD_PRELOAD=$D LSDHIP_LIB=$D LSDHIP_S3_TRACE=1 timeout 200 python tools/bench_sim3.py > $O/sim3_trace.out 2> $O/sim3_trace.err
sort $O/sim3_trace.err | uniq -c | sort -rn | head -12
cat $O/sim3_trace.out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o sim3 -- python $R/tools/bench_sim3.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for f in glob.glob(R + '/gpurun_out/r05_s11/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
