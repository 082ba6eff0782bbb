#!/usr/bin/env python
"""Developer check: the C++ track + map loop (liblsdhip_driver.so: next frame prefetched on the second stream, asynchronous mapping)
must give bit-identical poses and launch / evaluation counts run to run:  python tools/determinism_loop.py [reps] [frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lsd_slam_amd import synth
from lsd_slam_amd.driver import DriverLoop

w, h = 640, 480
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
frames, depth0, K, gt = synth.make_sequence(w, h, 41, seq_index=0)
dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
ptr = lambda i: dev[i % dev.shape[0]].data_ptr()
base, bad = None, 0
for r in range(reps):
    loop = DriverLoop(w, h, K, ptr(0), depth0, kf_every=10, images_on_device=True, device=0)
    sigs = []
    for chunk in range(0, n, 100):
        done, poses = loop.run([ptr(1 + chunk + k) for k in range(100)], want_poses=True)
        st = loop.stats()
        sigs.append((np.asarray(poses).tobytes(), int(st.track_launches), int(st.evaluations)))
    loop.close()
    sig = (b"".join(s[0] for s in sigs), [s[1] for s in sigs], [s[2] for s in sigs])
    if base is None:
        base = sig
        print("run 0: launches", sig[1], "evaluations", sig[2])
    elif sig != base:
        bad += 1
        a, b = np.frombuffer(base[0]).reshape(-1, 7), np.frombuffer(sig[0]).reshape(-1, 7)
        diff = (a != b).any(axis=1)
        print("run %d differs: poses equal %s (first differing frame %s), launches %s, evaluations %s" % (
            r, not diff.any(), int(np.argmax(diff)) if diff.any() else None, sig[1], sig[2]))
print("determinism (C++ loop): %d of %d repeat runs differ from the first" % (bad, reps - 1))
