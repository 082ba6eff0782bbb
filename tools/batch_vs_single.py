#!/usr/bin/env python
"""Developer check (round 6): lsdhip_tracker_track_batch in throughput mode (>= 8 jobs) against lsdhip_tracker_track, job by job — every result
field, not only the pose.  python tools/batch_vs_single.py [--jobs 8] [--size 640x480] [--frames 3]"""
import argparse, os, sys
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=8)
ap.add_argument("--size", default="640x480")
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--spec", type=int, default=-1)
args = ap.parse_args()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsd_slam_amd as hip
from lsd_slam_amd import synth
w, h = [int(v) for v in args.size.split("x")]
S = args.jobs
seqs = [synth.make_sequence(w, h, args.frames + 1, s % 8) for s in range(S)]
ctx = hip.Context(w, h, seqs[0][2], device=0)
refs, kfs = [], []
for s in range(S):
    kf = hip.Frame(ctx, 1000 * s, seqs[s][0][0]); kf.setDepthFromGroundTruth(seqs[s][1])
    r = hip.TrackingReference(); r.importFrame(kf); refs.append(r); kfs.append(kf)
trb, trs = hip.SE3Tracker(ctx), hip.SE3Tracker(ctx)
for t in (trb, trs):
    t.set_maxItsPerLvl([5, 20, 50, 100, 0])
    if args.spec > 0:
        t.set_speculation(args.spec)
inits = np.tile(hip.IDENTITY, (S, 1))
F = ("numEvaluations", "numWarpUpdates", "lastResidual", "pointUsage", "lastGoodCount", "lastBadCount", "lastMeanRes", "affineEstimation_a", "affineEstimation_b")
for t in range(1, args.frames + 1):
    frb = [hip.Frame(ctx, 1000 * s + t, seqs[s][0][t]) for s in range(S)]
    poses, res = trb.trackFrameBatch(refs, frb, inits)
    for s in range(S):
        fs = hip.Frame(ctx, 1000 * s + t, seqs[s][0][t])
        p = trs.trackFrame(refs[s], fs, inits[s])
        r = trs.last
        d = np.abs(np.asarray(p) - poses[s]).max()
        mask_b, mask_s = frb[s].refPixelWasGoodNoCreate(), fs.refPixelWasGoodNoCreate()
        ham = int((mask_b != mask_s).sum())
        print("frame %d job %d: |pose diff| %.1e, mask hamming %d; " % (t, s, d, ham) + ", ".join("%s %s/%s" % (k, getattr(res[s], k), getattr(r, k)) for k in F))
    inits = poses
