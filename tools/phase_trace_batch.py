#!/usr/bin/env python
"""Developer tool (round 6): where a fused round of a throughput-mode tracking batch spends its time.  LSD_PHASE_TRACE build
(lsd_slam_amd/build.py build_trace): one workgroup of job 0 (LSDHIP_TRACE_WG, default 0) leaves shader-clock timestamps at the phase
boundaries of k_track_step<256, true, TS_FUSED>.

    LSDHIP_LIB=lsd_slam_amd/liblsdhip_trace.so LD_PRELOAD=$LSDHIP_LIB LSDHIP_TRACE_FILE=/tmp/tr.txt python tools/phase_trace_batch.py --jobs 32
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lsd_slam_amd as la
from lsd_slam_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=32)
ap.add_argument("--rounds", type=int, default=6)
args = ap.parse_args()
w, h = 640, 480
B = args.jobs
torch.cuda.init()
base = [synth.make_sequence(w, h, 6, seq_index=s) for s in range(min(B, 8))]
ctx = la.Context(w, h, base[0][2])
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
for k in range(args.rounds):
    frs = [la.Frame(ctx, 1000 * j + k, device_ptr=dev[j][1 + k % 5].data_ptr()) for j in range(B)]
    tr.trackFrameBatch(refs, frs, inits)
ctx.synchronize()
tr.close()
path = os.environ.get("LSDHIP_TRACE_FILE")
if not (path and os.path.exists(path)):
    sys.exit("no trace file: LSDHIP_LIB must be the LSD_PHASE_TRACE build and LSDHIP_TRACE_FILE set")
a = np.loadtxt(path, dtype=np.float64)
full = a[(a[:, 6] > 0) & (a[:, 2] > 0) & (a[:, 21] > 0)]
wall = (full[:, 9] - full[:, 8]) * 10.0   # ns (100 MHz counter)
clk = full[:, 6] - full[:, 0]
ghz = np.median(clk / np.maximum(wall, 1))
print("traced workgroup %s of job 0; %d launches with all phases of %d traced; shader clock ~%.2f GHz; grid %d x %d" % (
    os.environ.get("LSDHIP_TRACE_WG", "0"), len(full), len(a), ghz, int(full[0, 24]), int(full[0, 25])))
order = [0, 1, 2, 3, 4, 20, 21, 22, 5, 6]
names = ["state load", "sums + tail + barrier", "totals", "LM step + barrier", "block loads + prefix", "list + barrier", "first two stages issued",
         "strip loop", "reduction + write"]
for lvl in sorted(set(full[:, 10].astype(int)), reverse=True):
    for same in (1, 0):
        m = (full[:, 10].astype(int) == lvl) & ((full[:, 26].astype(int) == lvl) == bool(same))
        if m.sum() == 0:
            continue
        f = full[m]
        d = np.diff(f[:, order], axis=1) / ghz / 1e3
        print("level %d (%s, %d launches, %d strips, list %d entries): body %.2f us = " % (
            lvl, "level unchanged" if same else "first evaluation of the level", m.sum(), int(f[0, 11]), int(np.median(f[:, 23])),
            np.median((f[:, 6] - f[:, 0]) / ghz / 1e3)) + ", ".join("%s %.2f" % (n, np.median(d[:, k])) for k, n in enumerate(names)))
        print("      inside sums + tail: sums branch done at %.2f us, tail branch at %.2f us after the state barrier" % (
            np.median((f[:, 17] - f[:, 1]) / ghz / 1e3), np.median((f[:, 18] - f[:, 1]) / ghz / 1e3)))
