#!/usr/bin/env python
"""Sim3Tracker::trackFrameSim3 timing on one MI355X (keyframe-to-keyframe constraint, levels 3..1 as the constraint search
uses them): ms per call, evaluations per call, µs per evaluation (kernel + finalize + host LM round trip).  Not the bench line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lsd_slam_amd as la
from lsd_slam_amd import synth

torch.cuda.init()
for (w, h) in ((640, 480), (1280, 1024)):
    sc = synth.Scene(0)
    K = synth.intrinsics(w, h)
    imgA, depthA = sc.render(0, w, h)
    imgB, depthB = sc.render(3, w, h)
    ctx = la.Context(w, h, K)
    a, b = la.Frame(ctx, 0, imgA), la.Frame(ctx, 3, imgB)
    a.setDepthFromGroundTruth(depthA)
    b.setDepthFromGroundTruth((depthB / 1.2).astype(np.float32))
    R, t = sc.frame_to_ref(3, 0)
    init = np.concatenate([synth.rot_to_quat(R), t, [1.0]])
    tr = la.Sim3Tracker(ctx)
    for _ in range(3):
        T, r = tr.trackFrameSim3(a, b, init, 3, 1)
    n, ev = 20, 0
    t0 = time.perf_counter()
    for _ in range(n):
        T, r = tr.trackFrameSim3(a, b, init, 3, 1)
        ev += r.numEvaluations
    dt = time.perf_counter() - t0
    # the same job eight times in one lock-step batch (lsdhip_sim3tracker_track_batch)
    B = 8
    for _ in range(2):
        tr.trackFrameSim3Batch([a] * B, [b] * B, np.tile(init, (B, 1)), 3, 1)
    tb = time.perf_counter()
    for _ in range(n):
        tr.trackFrameSim3Batch([a] * B, [b] * B, np.tile(init, (B, 1)), 3, 1)
    dtb = time.perf_counter() - tb
    print(json.dumps({"size": "%dx%d" % (w, h), "ms_per_job_in_batch_of_8": dtb / n / B * 1e3, "ms_per_trackFrameSim3": dt / n * 1e3, "evaluations_per_call": ev / n,
                      "us_per_evaluation": dt / ev * 1e6, "scale": float(T[7]), "diverged": bool(tr.diverged)}))
