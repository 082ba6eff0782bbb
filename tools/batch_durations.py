#!/usr/bin/env python
"""Launch durations of the batched k_track_step variants from a rocprofv3 kernel trace (developer tool)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_track_step" in r["Kernel_Name"] and "true" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def kind(n):
    return "LM" if "true, 1>" in n else ("EV" if "true, 2>" in n else "FU")
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -60:]
prev = None
out = []
for r in tail:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%s%.0f(g%d)" % (kind(r["Kernel_Name"]), (en - st) / 1e3, int(r["Grid_Size_X"]) // 256))
print(" ".join(out))
