#!/usr/bin/env python
"""Histogram of k_track_step<.., true> launch durations from a rocprofv3 kernel trace (developer tool)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_track_step" in r["Kernel_Name"] and "true" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = [int(r["Grid_Size_X"]) if "Grid_Size_X" in r else 0 for r in rows]
print("launches", len(d))
tail = d[-52:]
print("last 52 launches (us):", " ".join("%.0f" % x for x in tail))
print("grid x of those:", " ".join(str(x) for x in g[-52:]))
