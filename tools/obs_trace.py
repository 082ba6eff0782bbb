#!/usr/bin/env python
"""Per-lane timeline of k_observe from the LSD_PHASE_TRACE build (one traced lane per 32x8 tile, last observe call of the run).
Usage (GPU box): LSDHIP_LIB=lsd_slam_amd/liblsdhip_trace.so LSDHIP_OBS_TRACE_FILE=gpurun_out/x/obs.txt python tools/phase_trace.py;
                 python tools/obs_trace.py gpurun_out/x/obs.txt"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.float64)
ghz = 2.37
a = a[a[:, 0] > 0]
# (clock64 is per XCD: only differences within one lane are meaningful)
dur = (a[:, 7] - a[:, 0]) / ghz / 1e3
print("lane duration: p50 %.2f p90 %.2f p99 %.2f max %.2f us" % (np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
st = a[a[:, 5] > 0]
if len(st):
    walk = (st[:, 5] - st[:, 4]) / ghz / 1e3
    lc = st[:, 8]
    d2 = (st[:, 7] - st[:, 0]) / ghz / 1e3
    print("lanes that walked: %d; steps p50 %d p90 %d max %d; walk us p50 %.2f p90 %.2f max %.2f; whole lane p50 %.2f p90 %.2f max %.2f" % (
        len(st), np.median(lc), np.percentile(lc, 90), lc.max(), np.median(walk), np.percentile(walk, 90), walk.max(),
        np.median(d2), np.percentile(d2, 90), d2.max()))
    k = np.argmax(d2)
    print("slowest walking lane: stages (us)", [round((st[k, j] - st[k, 0]) / ghz / 1e3, 2) for j in (2, 3, 4, 5, 6, 7)], "steps", int(lc[k]))
