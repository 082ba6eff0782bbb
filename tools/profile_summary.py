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
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith("void k_track_step")]
    d.sort()
    work = [x for x in d if x > 5.5]   # launches queued behind the finishing one leave in 4-5 us (416 workgroups, one scalar load each)
    with open(os.path.join(dst, tag + "_track_step_durations.txt"), "w") as f:
        f.write("k_track_step launches: %d, mean %.2f us, median %.2f us\n" % (len(d), sum(d) / len(d), d[len(d) // 2]))
        f.write("launches > 5.5 us (did an evaluation): %d, mean %.2f us, median %.2f us, p90 %.2f us\n" %
                (len(work), sum(work) / len(work), work[len(work) // 2], work[int(len(work) * 0.9)]))
        f.write("launches <= 5.5 us (job already finished, early exit): %d\n" % (len(d) - len(work)))

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
