#!/bin/bash
# PMC counters of the throughput-mode evaluation launch (level 1, 64 jobs): tools/gpu_pmc_eval.sh <tag> "<counters>"
TAG=$1; CNT=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python $R/tools/bench_eval.py --levels 1 --repeats 5 > $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counters", open("gpurun_out/pmc_$TAG.log").read()[-1500:])
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_track_step" not in k: continue
        key = "EVAL" if "true, 2>" in k else ("LM" if "true, 1>" in k else "other")
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == list(agg[key].keys())[0]: n[key] += 1
    for key in agg:
        print(key, "dispatches", n[key], {c: "%.3g" % (v / max(1, n[key])) for c, v in agg[key].items()})
PY
rm -rf gpurun_out/pmc_$TAG
