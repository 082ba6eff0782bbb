#!/usr/bin/env python
"""Developer tool: per-phase timestamps of k_track_step (LSD_PHASE_TRACE build, lsd_slam_amd/build.py --trace).
Usage (GPU box): LSDHIP_LIB=lsd_slam_amd/liblsdhip_trace.so LSDHIP_TRACE_FILE=gpurun_out/x/trace.txt python tools/phase_trace.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import lsd_slam_amd as la  # noqa: E402
from lsd_slam_amd import synth  # noqa: E402

w, h = 640, 480
frames, depth0, K, gt = synth.make_sequence(w, h, 50)
d_frames = torch.from_numpy(frames).cuda(0)
ctx = la.Context(w, h, K, device=0)
loop = la.SlamLoop(ctx, d_frames[0].data_ptr(), depth0, kf_every=10, device_frames=True)
for i in range(1, 40):
    loop.step(d_frames[i % 50].data_ptr(), time.perf_counter)
ctx.synchronize()
loop.map.close()      # the observe trace (LSDHIP_OBS_TRACE_FILE) is written when the depth map goes away
loop.tracker.close()
path = os.environ.get("LSDHIP_TRACE_FILE")
if path and os.path.exists(path):
    a = np.loadtxt(path, dtype=np.float64)
    # columns: 0..6 phase clocks (shader clock), 8/9 wall clock (100 MHz) at start / end, 10 level, 11 nb
    full = a[(a[:, 6] > 0) & (a[:, 2] > 0)]
    names = ["load state", "sums+tail", "finalize", "LM", "residual loop", "reduce+write"]
    d = np.diff(full[:, 0:7], axis=1)
    wall = (full[:, 9] - full[:, 8]) * 10.0  # ns
    clk = (full[:, 6] - full[:, 0])
    ghz = np.median(clk / wall)
    print("launches with all phases: %d, shader clock ~%.2f GHz, kernel body %.2f us (median)" % (len(full), ghz, np.median(wall) / 1e3))
    if a.shape[1] >= 17:
        lm = full[(full[:, 16] > 0)]
        if len(lm):
            e = lm[:, [3, 12, 13, 14, 15, 16]]
            dd = np.diff(e, axis=1)
            print("LM (proposing launches, %d): " % len(lm) + ", ".join("%s %.2f us" % (n, np.median(dd[:, k]) / ghz / 1e3) for k, n in
                  enumerate(["epilogue", "decision+A/b", "matrix load", "LDLT", "exp/mul/store"])))
    if a.shape[1] >= 19:
        both = full[(full[:, 17] > 0) & (full[:, 18] > 0)]
        if len(both):
            print("inside sums+tail (from the state barrier): sums branch %.2f us, tail branch %.2f us (median, %d launches)" % (
                np.median(both[:, 17] - both[:, 1]) / ghz / 1e3, np.median(both[:, 18] - both[:, 1]) / ghz / 1e3, len(both)))
            for lvl in sorted(set(both[:, 10].astype(int))):
                m = both[:, 10].astype(int) == lvl
                print("   level %d: sums %.2f us, tail %.2f us" % (lvl, np.median(both[m][:, 17] - both[m][:, 1]) / ghz / 1e3,
                                                                  np.median(both[m][:, 18] - both[m][:, 1]) / ghz / 1e3))
    for lvl in sorted(set(full[:, 10].astype(int))):
        m = full[:, 10].astype(int) == lvl
        print("level %d (nb=%d, %d launches): " % (lvl, int(full[m][0, 11]), m.sum()) +
              ", ".join("%s %.2f us" % (n, np.median(d[m][:, k]) / ghz / 1e3) for k, n in enumerate(names)))

    if a.shape[1] >= 20:
        print("by (level evaluated, trials pending, trial the LM loop stopped at):")
        key = np.stack([full[:, 10], full[:, 19], full[:, 7]], axis=1).astype(int)
        for k in sorted(set(map(tuple, key))):
            m = (key == np.array(k)).all(axis=1)
            print("   L%d pending %d stop %d (%3d launches): " % (k[0], k[1], k[2], m.sum()) +
                  ", ".join("%s %.2f" % (n, np.median(d[m][:, j]) / ghz / 1e3) for j, n in enumerate(names)) +
                  " | body %.2f us" % (np.median(wall[m]) / 1e3))
