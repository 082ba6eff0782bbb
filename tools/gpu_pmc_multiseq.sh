#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes with --kernel-trace only, gfx950 corrections as the micro-architecture
# guide prescribes: counters in KiB, FETCH_SIZE x 2) of the shared launches of the 32-sequence loop, with the kernel durations of an
# unprofiled-counter trace: bytes per launch / duration = where those launches sit against the HBM roofline.  tools/gpu_pmc_multiseq.sh <tag>
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/multiseq_check.py --S 32 --steps 30"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mspmc_t -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/mspmc_f -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/mspmc_w -- $CMD > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, json, collections
def counters(d, name):
    f = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    agg, n = collections.defaultdict(float), collections.Counter()
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"].split("(")[0]
            agg[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: agg[k] / n[k] for k in agg}
fetch, write = counters("mspmc_f", "FETCH_SIZE"), counters("mspmc_w", "WRITE_SIZE")
st = glob.glob("gpurun_out/mspmc_t/**/*kernel_stats.csv", recursive=True)
dur = {r["Name"].split("(")[0]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(st[0]))} if st else {}
out = {"command": "tools/multiseq_check.py --S 32 --steps 30 (32 sequences x 640x480, blockUntilMapped batch loop)",
       "correction": "FETCH_SIZE x 2 (gfx950), both counters in KiB", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if "batch" not in k and "k_track_step<256, true" not in k: continue
    rb, wb = 2.0 * 1024 * fetch.get(k, 0.0), 1024 * write.get(k, 0.0)
    rec = {"read_bytes_per_launch": rb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": rb + wb}
    if k in dur:
        us = dur[k][0] / 1e3
        rec.update({"avg_launch_us": us, "launches": dur[k][1], "hbm_GBps": (rb + wb) / (us * 1e-6) / 1e9, "frac_of_8TBps": (rb + wb) / (us * 1e-6) / 8e12})
    out["kernels"][k] = rec
json.dump(out, open("gpurun_out/%s_multiseq_pmc.json" % "$TAG", "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-44s %8.1f MB/launch %7.1f us  %6.2f TB/s (%4.1f %%)" % (k[:44], v["hbm_bytes_per_launch"] / 1e6, v.get("avg_launch_us", 0), v.get("hbm_GBps", 0) / 1e3, 100 * v.get("frac_of_8TBps", 0)))
PY
rm -rf gpurun_out/mspmc_t gpurun_out/mspmc_f gpurun_out/mspmc_w
