#!/usr/bin/env python
"""Developer tool: run-to-run determinism of the pipelined C++ loop (poses of `n` frames, `reps` runs); prints the first differing frame.
LSDHIP_NO_PREFETCH=1 = no next-frame prefetch in the enqueue hook.  (The LSDHIP_PIPE_DEBUG bisection switches of the round-4 bug hunt —
profiles/r04_notes.md section 1a — have left the library with the bug.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import sequence
from lsd_slam_amd.driver import DriverLoop
w, h = 640, 480
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 1
frames, depth0, K, gt = sequence(w, h, 41)
dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
ptr = lambda i: dev[i % dev.shape[0]].data_ptr()
base = None
firsts = []
for rep in range(reps):
    drv = DriverLoop(w, h, K, ptr(0), depth0, kf_every=10, images_on_device=True)
    if pipe:
        drv.set_pipeline(True)
    done, poses = drv.run([ptr(1 + k) for k in range(n)], want_poses=True)
    drv.close()
    poses = np.asarray(poses)
    if base is None:
        base = poses
    else:
        d = np.where(np.any(poses != base, axis=1))[0]
        firsts.append(int(d[0]) + 1 if len(d) else 0)
print("NO_PREFETCH=%s pipe=%d: first differing frame per run (0 = identical): %s" % (os.environ.get("LSDHIP_NO_PREFETCH", "0"), pipe, firsts))
