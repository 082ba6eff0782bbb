#!/usr/bin/env python
"""Developer tool: the launch timeline of the throughput-mode tracking batches out of a rocprofv3 kernel trace (kernel_trace.csv): per batch the
rounds, per round LM duration / gap / evaluation duration / gap, and the totals.   python tools/batch_timeline.py <kernel_trace.csv> [jobs]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tr = [r for r in rows if "k_track_step<256, true" in r["Kernel_Name"] and int(r["Grid_Size_Y"]) == jobs]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(("LM" if "true, 1>" in r["Kernel_Name"] else ("EV" if "true, 2>" in r["Kernel_Name"] else "FU")), int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3, int(r["Grid_Size_X"])) for r in tr]
# batches: a gap of more than 60 us between tracking launches
batches, cur = [], []
for e in ev:
    if cur and e[1] - cur[-1][2] > 60.0:
        batches.append(cur); cur = []
    cur.append(e)
if cur:
    batches.append(cur)
print("%d tracking launches with %d jobs in %d batches" % (len(ev), jobs, len(batches)))
tot = {"LM": 0.0, "EV": 0.0, "FU": 0.0, "gap": 0.0, "span": 0.0, "n": 0, "idle_tail": 0.0}
hist = {}
for b in batches[len(batches) // 3:]:
    # launches behind the last job's finishing step are ~2-3 us evaluations: the working part ends at the last evaluation longer than 4 us
    last = max([i for i, e in enumerate(b) if e[0] != "LM" and e[2] - e[1] > 4.0] + [0])
    w = b[:last + 1]
    tot["span"] += w[-1][2] - w[0][1]
    tot["idle_tail"] += b[-1][2] - w[-1][2]
    tot["n"] += 1
    for i, e in enumerate(w):
        tot[e[0]] += e[2] - e[1]
        if i:
            tot["gap"] += e[1] - w[i - 1][2]
        if e[0] != "LM":
            k = int((e[2] - e[1]) // 4) * 4
            hist[k] = hist.get(k, 0) + 1
n = max(1, tot["n"])
print("per batch (last two thirds, %d batches): span %.1f us = LM %.1f + evaluation %.1f + fused %.1f + gaps %.1f; launches queued behind the finishing step %.1f us" % (
    n, tot["span"] / n, tot["LM"] / n, tot["EV"] / n, tot["FU"] / n, tot["gap"] / n, tot["idle_tail"] / n))
print("evaluation launch durations (us bucket: count per batch): " + ", ".join("%d: %.1f" % (k, hist[k] / n) for k in sorted(hist)))
b = batches[-2] if len(batches) > 1 else batches[-1]
t0 = b[0][1]
print("one batch: " + " ".join("%s%d:%.0f@%.0f" % (e[0], e[3] // 256, e[2] - e[1], e[1] - t0) for e in b[:70]))
