#!/usr/bin/env python
"""Throughput-mode residual evaluation launch alone (lsdhip_tracker_eval_throughput): J jobs of a 640x480 (or WxH) frame pair per
launch at pyramid levels 3, 2, 1; mean launch time over R identical launches, algorithmic GB/s and fraction of the 8 TB/s HBM peak.
Usage: python tools/bench_eval.py [--jobs 64] [--size 640x480] [--repeats 20]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lsd_slam_amd as la
from lsd_slam_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=64)
ap.add_argument("--size", type=str, default="640x480")
ap.add_argument("--repeats", type=int, default=20)
ap.add_argument("--levels", type=str, default="3,2,1")
args = ap.parse_args()
w, h = (int(v) for v in args.size.split("x"))
frames, depth0, K, gt = synth.make_sequence(w, h, 6)
d_frames = torch.from_numpy(frames).cuda()
ctx = la.Context(w, h, K)
refs, keep = [], []
for j in range(args.jobs):
    kf = la.Frame(ctx, 1000 + j, device_ptr=d_frames[0].data_ptr())
    kf.setDepthFromGroundTruth(depth0)
    r = la.TrackingReference()
    r.importFrame(kf)
    refs.append(r)
    keep.append(kf)
frs = [la.Frame(ctx, 2000 + j, device_ptr=d_frames[1 + j % 5].data_ptr()) for j in range(args.jobs)]
tr = la.SE3Tracker(ctx)
# poses: the ground-truth relative poses (reference -> frame), i.e. where the LM loop ends up
T = np.zeros((args.jobs, 7), np.float32)
for j in range(args.jobs):
    f2r = gt[1 + j % 5]            # frame -> reference (q, t)
    q, t = np.array(f2r[:4]), np.array(f2r[4:7])
    qi = np.array([q[0], -q[1], -q[2], -q[3]])
    # rotate -t by q^-1
    R = la.quat_to_rot(qi) if hasattr(la, "quat_to_rot") else None
    if R is None:
        ww, x, y, z = qi
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                      [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                      [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])
    T[j, :4] = qi
    T[j, 4:] = -(R @ t)
out = []
for lvl in [int(v) for v in args.levels.split(",")]:
    ms, nb = tr.evalThroughput(refs, frs, T, lvl, args.repeats)
    out.append({"level": lvl, "jobs": args.jobs, "us_per_launch": ms * 1e3, "algorithmic_MB_per_launch": nb / 1e6,
                "achieved_GBps": nb / (ms * 1e-3) / 1e9, "frac_of_8TBps": nb / (ms * 1e-3) / 8e12})
    print(json.dumps(out[-1]))
