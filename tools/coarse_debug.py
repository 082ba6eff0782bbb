#!/usr/bin/env python
"""Developer tool: the cluster kernel (k_track_coarse) against the k_track_step chain and the oracle, frame by frame, and —
with the LSD_PHASE_TRACE build (LSDHIP_LIB=lsd_slam_amd/liblsdhip_trace.so LSDHIP_CTRACE_FILE=...) — its per-step phase times."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lsd_slam_amd as hip  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import test_gpu_parity as T  # noqa: E402

oracle.build()
sizes = [(176, 144), (640, 480)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (w, h) in sizes:
    frames, depth0, K, gt, ctx = T.make_pair(oracle, hip, w, h, 6)
    kfo, kfg = oracle.Frame(0, frames[0], K), hip.Frame(ctx, 0, frames[0])
    kfo.set_depth_gt(depth0)
    kfg.setDepthFromGroundTruth(depth0)
    ro, rg = oracle.TrackingReference(), hip.TrackingReference()
    ro.import_frame(kfo)
    rg.importFrame(kfg)
    tro = oracle.SE3Tracker(w, h, K, mode=oracle.SSE)
    tro.set_max_its(T.ODOMETRY_ITS)
    variants = [("chain t1", 0, 1, 0), ("chain def", 0, 0, 0), ("coarse t1", 1, 1, 0), ("coarse nofold", 1, 0, 0), ("coarse fold", 1, 0, 1)]
    trs = []
    for name, coarse, t1, fold in variants:
        tr = hip.SE3Tracker(ctx)
        tr.set_maxItsPerLvl(T.ODOMETRY_ITS)
        tr.set_coarse(bool(coarse), bool(fold))
        if t1:
            tr.set_speculation(1)
        trs.append(tr)
    init = T.IDENT7.copy()
    for i in range(1, 6):
        r = tro.track(ro, oracle.Frame(i, frames[i], K), init)
        po = np.array(r.frameToRef)
        print("%dx%d frame %d oracle: evals %d affine (%.6f %.4f) residual %.6f" % (w, h, i, r.numEvaluations, r.affine_a, r.affine_b, r.lastResidual))
        for (name, *_), tr in zip(variants, trs):
            est = tr.trackFrame(rg, hip.Frame(ctx, i, frames[i]), init)
            dt, dr = T.pose_distance(est, po, oracle)
            print("   %-14s evals %2d levels %s steps %s affine (%.6f %.4f) residual %.6f |dpose| %.1e %.1e" % (
                name, tr.last.numEvaluations, tr.exec_stats()[3], tr.step_stats()[:3], tr.affineEstimation_a, tr.affineEstimation_b,
                tr.lastResidual, dt, dr))
        init = po
    for tr in trs:
        tr.close()
path = os.environ.get("LSDHIP_CTRACE_FILE")
if path and os.path.exists(path):
    a = np.loadtxt(path, dtype=np.float64)
    a = a[a[:, 4] > 0]
    ghz = 2.4
    print("cluster kernel steps traced: %d" % len(a))
    names = ["eval+reduce", "publish", "gather+sum", "scan+tail", "LM"]
    for lvl in sorted(set(a[:, 5].astype(int))):
        for nc in sorted(set(a[a[:, 5] == lvl][:, 6].astype(int))):
            m = (a[:, 5] == lvl) & (a[:, 6] == nc) & (a[:, 1] > 0)
            if m.sum() == 0:
                continue
            d = np.diff(a[m][:, [0, 1, 2, 3, 12, 4]], axis=1)
            print("  level %d trials %d strips %d (%3d steps): " % (lvl, nc, int(np.median(a[m][:, 7])), m.sum()) +
                  ", ".join("%s %.2f" % (n, np.median(d[:, k]) / ghz / 1e3) for k, n in enumerate(names)) + " | step %.2f us (at %.1f GHz)" % (np.median(d.sum(axis=1)) / ghz / 1e3, ghz))
    first = a[a[:, 11] == 0]
    if len(first):
        print("  point load (kernel entry -> first step): %.2f us median" % (np.median(first[:, 10]) / ghz / 1e3))
