#!/usr/bin/env python
"""Developer tool: S replicated sequences through lsd_slam_hip::SlamLoopBatch, twice — which sequences differ from their replica
(same inputs, another z-slice of the launches) and which differ run to run, and from which step on.
  python tools/multiseq_check.py [--S 32] [--steps 20]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, nargs="+", default=[8, 16, 32])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--motions", type=int, default=4)
    args = ap.parse_args()
    import torch
    from lsd_slam_amd import synth
    from lsd_slam_amd.driver import DriverLoopBatch
    w, h = 640, 480
    frames, depth0, K, gt = synth.make_sequence(w, h, 50, seq_index=0, kind="s1")
    d = torch.from_numpy(frames).cuda(0)
    n = d.shape[0]
    motions = [(1, 1), (-1, 1), (1, 2), (-1, 2)][:args.motions]
    M = len(motions)
    for S in args.S:
        runs = []
        for rep in range(2):
            idx = lambda s, t: (motions[s % M][0] * motions[s % M][1] * t) % n
            ptrs = lambda t: [d[idx(s, t)].data_ptr() for s in range(S)]
            loop = DriverLoopBatch(w, h, K, ptrs(0), [depth0] * S, kf_every=10, images_on_device=True, device=0)
            done, poses = loop.run([ptrs(1 + k) for k in range(args.steps)], want_poses=True)
            st = loop.stats()
            loop.close()
            runs.append(poses)
        for rep in range(2):
            bad = []
            for s in range(S):
                ne = np.any(runs[rep][:, s] != runs[rep][:, s % M], axis=1)
                if ne.any():
                    bad.append((s, int(np.argmax(ne)), float(np.abs(runs[rep][:, s] - runs[rep][:, s % M]).max())))
            print("S=%d run %d: %d sequences differ from their replica: %s" % (S, rep, len(bad), bad[:8]), flush=True)
        ne = np.any(runs[0] != runs[1], axis=2)
        print("S=%d run 0 vs run 1: %d of %d sequences differ, first steps %s" % (S, int(ne.any(axis=0).sum()), S,
              [int(np.argmax(ne[:, s])) for s in range(S) if ne[:, s].any()][:8]), flush=True)


if __name__ == "__main__":
    main()
