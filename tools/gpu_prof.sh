#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py run: tools/gpu_prof.sh <tag> [bench flags]   -> gpurun_out/<tag>_kernel_stats.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline-events --no-throughput-mode --no-pcie-leg "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
cd $R
python - <<PY
import csv, glob, shutil
f = glob.glob("gpurun_out/prof_$TAG/**/*kernel_stats.csv", recursive=True)
if f:
    shutil.copy(f[0], "gpurun_out/${TAG}_kernel_stats.csv")
    for r in list(csv.DictReader(open(f[0])))[:12]:
        print("%-70s n=%5s avg=%8.2f us tot=%9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"][:5]))
else:
    print("no stats", open("gpurun_out/prof_$TAG.err").read()[-1500:])
PY
rm -rf gpurun_out/prof_$TAG
