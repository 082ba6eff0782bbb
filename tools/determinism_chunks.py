#!/usr/bin/env python
"""Developer check: the C++ track + map loop run `reps` times over the same frames, statistics read every `chunk` frames — which chunk of
which run differs from run 0 in poses / launches / evaluations?   python tools/determinism_chunks.py [--reps 10] [--frames 150] [--chunk 10] [--pipelined]
(works from any checkout: it imports the package next to the tools directory it lives in, or of --root)"""
import argparse, os, sys
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--frames", type=int, default=150)
ap.add_argument("--chunk", type=int, default=10)
ap.add_argument("--pipelined", action="store_true")
ap.add_argument("--root", type=str, default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = ap.parse_args()
sys.path.insert(0, args.root)
import torch
from lsd_slam_amd import synth
from lsd_slam_amd.driver import DriverLoop
w, h = 640, 480
frames, depth0, K, gt = synth.make_sequence(w, h, 41, seq_index=0)
dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
ptr = lambda i: dev[i % dev.shape[0]].data_ptr()
base, bad = None, 0
for r in range(args.reps):
    loop = DriverLoop(w, h, K, ptr(0), depth0, kf_every=10, images_on_device=True, device=0)
    if args.pipelined:
        loop.set_pipeline(True)
    sigs = []
    for c0 in range(0, args.frames, args.chunk):
        done, poses = loop.run([ptr(1 + c0 + k) for k in range(args.chunk)], want_poses=True)
        st = loop.stats()
        sigs.append((np.asarray(poses).copy(), int(st.track_launches), int(st.evaluations)))
    loop.close()
    if base is None:
        base = sigs
        continue
    notes = []
    for i, (a, b) in enumerate(zip(base, sigs)):
        pe = np.array_equal(a[0], b[0])
        if not pe or a[1] != b[1] or a[2] != b[2]:
            notes.append("chunk %d: poses %s (max |d| %.2e), launches %d vs %d, evaluations %d vs %d" % (
                i, "equal" if pe else "DIFFER", float(np.abs(a[0] - b[0]).max()), a[1], b[1], a[2], b[2]))
    if notes:
        bad += 1
        print("run %d differs from run 0: %s" % (r, "; ".join(notes[:4])))
print("%s pipelined=%s: %d of %d repeat runs differ from the first" % (args.root, args.pipelined, bad, args.reps - 1))
