#!/usr/bin/env python
"""Print the kernel timeline of a few frames from a rocprofv3 kernel trace csv (developer tool)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frame boundary = k_image_pyramid
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_image_pyramid")]
for fi in (30, 31):
    a, b = starts[fi], starts[fi + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev = None
    n_step = 0
    for r in rows[a:b]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][:40]
        gap = (st - prev) / 1e3 if prev else 0.0
        if "k_track_step" in name:
            n_step += 1
            if n_step > 2 and gap < 3.0 and "--all" not in sys.argv:
                prev = en
                continue
        q = r.get("Queue_Id", "?")   # two queues on a pipelined context: tracking stream / mapping stream
        print("q=%-3s %-42s start=%8.1f end=%8.1f dur=%6.1f gap=%6.1f" % (q, name, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, gap))
        prev = en
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3
    print("frame total %.1f us, %d track steps, kernels busy %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, n_step, busy))
    print("----")
