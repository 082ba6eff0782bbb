#!/usr/bin/env python
"""Developer check: the HIP track+map loop must give bit-identical results run to run (python tools/determinism_check.py [reps])."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lsd_slam_amd as la
from common import sequence
import seq_loops

w, h, n = 640, 480, 50
frames, depth0, K, gt = sequence(w, h, n + 1, 0, "S1")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
base = None
bad = 0
for r in range(reps):
    ctx = la.Context(w, h, K)
    rec = seq_loops.run_hip(la, ctx, frames, depth0, n, 10)
    sig = (np.array(rec.frameToKF).tobytes(), np.array(rec.residual, np.float32).tobytes(), rec.final_map["idepth"].tobytes(), rec.final_map["isValid"].tobytes())
    if base is None:
        base = sig
    else:
        same = [a == b for a, b in zip(base, sig)]
        if not all(same):
            bad += 1
            f2k = np.frombuffer(sig[0]).reshape(-1, 7); f0 = np.frombuffer(base[0]).reshape(-1, 7)
            first = int(np.argmax((f2k != f0).any(axis=1)))
            print("run %d differs: poses %s residuals %s idepth %s valid %s; first differing frame %d, max |dpose| %.3g" % (r, same[0], same[1], same[2], same[3], first, np.abs(f2k - f0).max()))
print("determinism: %d of %d repeat runs differ from the first" % (bad, reps - 1))
