#!/usr/bin/env python
"""Summarise a tools/gpu_profile.sh output directory into the small files that get committed under profiles/:
  <tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (copied verbatim)
  <tag>_track_step_by_grid.txt  duration statistics of k_track_step split by launch kind (working / idle launches)
  <tag>_pmc_traffic.json     FETCH_SIZE / WRITE_SIZE per launch of every kernel, with the gfx950 correction applied
                             (FETCH_SIZE x 2 for wide coalesced reads; the counters are reported in KiB)."""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    g = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return g[0] if g else None


ks = find("trace", "*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, tag + "_kernel_stats.csv"))
kt = find("trace", "*kernel_trace.csv")
if kt:
    rows = list(csv.DictReader(open(kt)))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith("void k_track_step<256, false, 0>")]   # the single-sequence chain (the bench's extra legs run batch forms too)
    d.sort()
    work = [x for x in d if x > 5.5]   # launches queued behind the finishing one leave in 4-5 us (416 workgroups, one scalar load each)
    with open(os.path.join(dst, tag + "_track_step_durations.txt"), "w") as f:
        f.write("k_track_step launches: %d, mean %.2f us, median %.2f us\n" % (len(d), sum(d) / len(d), d[len(d) // 2]))
        f.write("launches > 5.5 us (did an evaluation): %d, mean %.2f us, median %.2f us, p90 %.2f us\n" %
                (len(work), sum(work) / len(work), work[len(work) // 2], work[int(len(work) * 0.9)]))
        f.write("launches <= 5.5 us (job already finished, early exit): %d\n" % (len(d) - len(work)))

# throughput-mode evaluation launches (tools/bench_eval.py --levels 1 under rocprofv3 --kernel-trace): per-launch durations
et = find("eval_trace", "*kernel_trace.csv")
if et:
    rows = [r for r in csv.DictReader(open(et)) if "k_track_step" in r["Kernel_Name"] and "true, 2>" in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    info = {}
    try:
        info = json.loads(open(os.path.join(out, "eval_l1.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    with open(os.path.join(dst, tag + "_eval_launches.txt"), "w") as f:
        f.write("k_track_step<256, true, TS_EVAL> launches at level 1, 64 jobs (tools/bench_eval.py --levels 1 --repeats 20 under rocprofv3 --kernel-trace)\n")
        if d:
            mb = info.get("algorithmic_MB_per_launch")
            f.write("launches: %d, durations us: %s\n" % (len(d), " ".join("%.1f" % x for x in d)))
            f.write("mean %.2f us, median %.2f us\n" % (sum(d) / len(d), sorted(d)[len(d) // 2]))
            if mb:
                f.write("algorithmic bytes per launch %.3f MB -> %.0f GB/s at the median duration = %.1f %% of the 8 TB/s HBM peak\n" %
                        (mb, mb * 1e6 / (sorted(d)[len(d) // 2] * 1e-6) / 1e9, mb * 1e6 / (sorted(d)[len(d) // 2] * 1e-6) / 8e12 * 100))
        if info:
            f.write("HIP-event figure of the same run (profiled): %s\n" % json.dumps(info))
for fn in ("eval_levels.json",):
    pth = os.path.join(out, fn)
    if os.path.exists(pth):
        shutil.copy(pth, os.path.join(dst, tag + "_" + fn))

traffic = {}
for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    acc = {}
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(k, [0, 0.0, set()])
        did = r.get("Dispatch_Id")
        if did not in a[2]:
            a[2].add(did)
            a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k, (n, v, _) in acc.items():
        traffic.setdefault(k, {"launches": n})[name + "_KiB_per_launch_raw"] = v / max(1, n)
for k, t in traffic.items():
    f = t.get("FETCH_SIZE_KiB_per_launch_raw")
    w = t.get("WRITE_SIZE_KiB_per_launch_raw")
    t["read_bytes_per_launch"] = None if f is None else f * 1024.0 * 2.0   # gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads
    t["write_bytes_per_launch"] = None if w is None else w * 1024.0
    t["hbm_bytes_per_launch"] = None if f is None or w is None else t["read_bytes_per_launch"] + t["write_bytes_per_launch"]
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB units; FETCH_SIZE doubled (gfx950, "
                   "MI355X_MICROARCH.md §HBM); Infinity-Cache hits are counted by these memory-side counters",
           "kernels": traffic}, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for fn in ("bench.json", "bench_under_trace.json"):
    p = os.path.join(out, fn)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, tag + "_" + fn))
print("summary files:", sorted(os.listdir(dst)))
