#!/usr/bin/env python
"""Throughput mode of the tracker: B independent 640x480 sequences tracked in the same launches on one MI355X
(lsdhip_tracker_track_batch, one job per blockIdx.y).  Tracking only (frame upload + pyramids + trackFrame), keyframe =
frame 0 of every sequence with ground-truth depth.  Prints one JSON line per batch size: aggregate frames/s and the
achieved algorithmic bandwidth of k_track_step (all jobs of a launch counted).  Not the driver's bench line."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lsd_slam_amd as la
from lsd_slam_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=str, default="1,4,16,32")
ap.add_argument("--rounds", type=int, default=20)
args = ap.parse_args()
w, h = 640, 480
Bmax = max(int(b) for b in args.batches.split(","))
torch.cuda.init()
base = [synth.make_sequence(w, h, 6, seq_index=s) for s in range(min(Bmax, 8))]
ctx = la.Context(w, h, base[0][2])
for B in [int(b) for b in args.batches.split(",")]:
    refs, kfs = [], []
    for j in range(B):
        frames, depth0, K, gt = base[j % len(base)]
        kf = la.Frame(ctx, 1000 * j, frames[0])
        kf.setDepthFromGroundTruth(depth0)
        r = la.TrackingReference()
        r.importFrame(kf)
        refs.append(r)
        kfs.append(kf)
    tr = la.SE3Tracker(ctx)
    tr.set_maxItsPerLvl([5, 20, 50, 100, 0])
    dev = [torch.from_numpy(base[j % len(base)][0]).cuda() for j in range(B)]
    inits = np.tile(la.IDENTITY, (B, 1))
    def one_round(k):
        frs = [la.Frame(ctx, 1000 * j + k, device_ptr=dev[j][1 + k % 5].data_ptr()) for j in range(B)]
        poses, recs = tr.trackFrameBatch(refs, frs, inits)
        return recs
    for k in range(3):
        one_round(k)
    ctx.prof_reset(); ctx.prof_enable(True); ctx.synchronize()
    t0 = time.perf_counter()
    ev = 0
    for k in range(args.rounds):
        ev += sum(r.numEvaluations for r in one_round(k))
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    ms, n_eval, nbytes = ctx.prof_read()
    # n_eval evaluations were spread over launches that each carried up to B jobs: launches ~ max evaluations of a job per round
    print(json.dumps({"batch": B, "frames_per_s": B * args.rounds / dt, "evaluations": int(n_eval),
                      "stream_ms_in_track_steps": ms, "algorithmic_GB": nbytes / 1e9,
                      "achieved_GBps": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None,
                      "frac_of_8TBps": nbytes / (ms * 1e-3) / 8e12 if ms > 0 else None}))
    del tr, refs, kfs
