#!/usr/bin/env python
"""Developer check: the one-stream C++ loop run `reps` times over the same frames, one lsdloop_run call per frame — which FRAME of which run
differs from run 0 in launches / evaluations per level (poses are compared too)?   python tools/determinism_frames.py [--reps 8] [--frames 150]"""
import argparse, os, sys
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--frames", type=int, default=150)
ap.add_argument("--pipelined", action="store_true")
args = ap.parse_args()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsd_slam_amd import synth
from lsd_slam_amd.driver import DriverLoop
w, h = 640, 480
frames, depth0, K, gt = synth.make_sequence(w, h, 41, seq_index=0)
dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
ptr = lambda i: dev[i % dev.shape[0]].data_ptr()
runs = []
for r in range(args.reps):
    loop = DriverLoop(w, h, K, ptr(0), depth0, kf_every=10, images_on_device=True, device=0)
    if args.pipelined:
        loop.set_pipeline(True)
    rows, prev = [], (0, 0, [0] * 5, 0)
    for f in range(args.frames):
        done, poses = loop.run([ptr(1 + f)], want_poses=True)
        st = loop.stats()
        cur = (int(st.track_launches), int(st.evaluations), [int(v) for v in st.level_evaluations], int(st.keyframes))
        rows.append((cur[0] - prev[0], cur[1] - prev[1], tuple(a - b for a, b in zip(cur[2], prev[2])), cur[3] - prev[3], np.asarray(poses).tobytes()))
        prev = cur
    loop.close()
    runs.append(rows)
bad = 0
for r in range(1, args.reps):
    for f, (a, b) in enumerate(zip(runs[0], runs[r])):
        if a[:4] != b[:4] or a[4] != b[4]:
            bad += 1
            print("run %d frame %d: launches %d vs %d, evaluations %d vs %d, per level %s vs %s, keyframe %d, poses %s" % (
                r, f, a[0], b[0], a[1], b[1], a[2], b[2], a[3], "equal" if a[4] == b[4] else "DIFFER"))
print("pipelined=%s: %d (run, frame) pairs differ from run 0; launches per frame of run 0: %s" % (args.pipelined, bad, [x[0] for x in runs[0][:30]]))
