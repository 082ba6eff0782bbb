#!/usr/bin/env python
"""Developer tool: stage timestamps of k_observe (LSD_PHASE_TRACE build): one traced lane (thread 0 = first compacted
candidate) per workgroup, from the last updateKeyframe of a short run."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsd_slam_amd as la
from lsd_slam_amd import synth
w, h = 640, 480
frames, depth0, K, gt = synth.make_sequence(w, h, 12)
ctx = la.Context(w, h, K)
loop = la.SlamLoop(ctx, frames[0], depth0, kf_every=10)
for i in range(1, 8):
    loop.step(frames[i], time.perf_counter)
ctx.synchronize()
loop.map.close()
a = np.loadtxt(os.environ["LSDHIP_OBS_TRACE_FILE"], dtype=np.float64)
ghz = 2.37
full = a[(a[:, 6] > 0) & (a[:, 7] > 0)]
print("workgroups traced through the whole stereo: %d of %d" % (len(full), len(a)))
names = ["precheck+compaction", "select..KF samples start", "KF samples", "ref samples", "walk", "post (subpixel, depth)", "EKF update + stores"]
d = np.diff(full[:, 0:8], axis=1)
print(", ".join("%s %.2f us" % (n, np.median(d[:, k]) / ghz / 1e3) for k, n in enumerate(names)))
print("walk steps: median %d, max %d; whole kernel body median %.2f us, p95 %.2f us" % (
    np.median(full[:, 8]), full[:, 8].max(), np.median(full[:, 7] - full[:, 0]) / ghz / 1e3, np.percentile(full[:, 7] - full[:, 0], 95) / ghz / 1e3))
per_step = (full[:, 5] - full[:, 4]) / np.maximum(full[:, 8], 1) / ghz
print("ns per walk step: median %.0f" % np.median(per_step))
