#!/usr/bin/env python
"""Developer check of the persistent tracking kernel: the same trackFrame jobs with LSDHIP_PERSIST on / off — poses, flags,
masks, execution statistics and wall time per call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsd_slam_amd as la  # noqa: E402
from lsd_slam_amd import synth  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
frames, depth0, K, gt = synth.make_sequence(w, h, 8)
res = {}
MODES = sys.argv[3].split(",") if len(sys.argv) > 3 else ["128", "0"]
for mode in MODES:

    ctx = la.Context(w, h, K)
    kf = la.Frame(ctx, 0, frames[0])
    kf.setDepthFromGroundTruth(depth0)
    ref = la.TrackingReference()
    ref.importFrame(kf)
    tr = la.SE3Tracker(ctx)
    tr.set_persistent(int(mode))
    tr.set_maxItsPerLvl([5, 20, 50, 100, 0])
    out = []
    init = la.IDENTITY.copy()
    for rep in range(3):
        init = la.IDENTITY.copy()
        for i in range(1, 8):
            f = la.Frame(ctx, i, frames[i])
            ctx.synchronize()
            t0 = time.perf_counter()
            est = tr.trackFrame(ref, f, init)
            dt = time.perf_counter() - t0
            out.append((est.copy(), tr.diverged, tr.trackingWasGood, tr.last.numEvaluations, tr.lastResidual, tr.pointUsage,
                        f.refPixelWasGoodNoCreate().copy(), dt, tr.exec_stats()))
            init = est
    if mode != "0":
        ph = np.zeros(10, np.uint64)
        tr.L.lsdhip_tracker_phase_cycles(tr.h_, ph.ctypes.data)
        n = tr.last.numEvaluations
        names = ["level entry", "warp+issue", "wg top3", "residual", "reduce+publish", "gather", "totals+tail", "LM"]
        print("phase cycles of the last job (%d evaluations, total %d): " % (n, ph.sum()) + ", ".join("%s %.0f" % (nm, v / n) for nm, v in zip(names, ph[:8])) + " per evaluation; gather cycles per evaluation at level 1: %.0f, level 3: %.0f" % (ph[8] / max(1, tr.exec_stats()[3][1]), ph[9] / max(1, tr.exec_stats()[3][3])))
    res[mode] = out
    print("PERSIST=%s exec_stats %s" % (mode, out[-1][8]))
for m in MODES:
    us = np.array([o[7] for o in res[m]][7:]) * 1e6
    ev = np.array([o[3] for o in res[m]][7:])
    print("mode %s: mean %.1f us per trackFrame, %.2f us per evaluation" % (m, us.mean(), us.sum() / ev.sum()))
for i, (a, b) in enumerate(zip(res[MODES[0]], res[MODES[-1]])):
    dp = np.abs(a[0] - b[0]).max()
    print("job %2d: |dpose| %.2e flags %s/%s evals %d/%d res %.5f/%.5f usage %.4f/%.4f mask diff %.2e  us %.0f / %.0f  levels %s"
          % (i, dp, (a[1], a[2]), (b[1], b[2]), a[3], b[3], a[4], b[4], a[5], b[5], (a[6] != b[6]).mean(), a[7] * 1e6, b[7] * 1e6, a[8][3]))
