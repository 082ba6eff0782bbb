#!/bin/bash
# S-sequence loop: blockUntilMapped (keyframe changes on side streams) against pipelined (tracking batch beside the mapping work, mapper one step behind)
cd "$(dirname "$0")/.."
python -m pytest tests/test_multiseq_gpu.py -x -q 2>&1 | tail -3
python - <<'PY' 2>/dev/null
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from lsd_slam_amd import synth, capi
from lsd_slam_amd.driver import DriverLoopBatch
w, h = 640, 480
frames, depth0, K, gt = synth.make_sequence(w, h, 60, seq_index=0, kind="s1")
d = torch.from_numpy(frames).cuda(0)
n = d.shape[0]
motions = [(1, 1), (-1, 1), (1, 2), (-1, 2)]
L = capi.lib()
for rep in range(2):
    for S in (8, 32):
        for overlap in (0, 1):
            idx = lambda s, t: (motions[s % 4][0] * motions[s % 4][1] * t) % n
            ptrs = lambda t: [d[idx(s, t)].data_ptr() for s in range(S)]
            loop = DriverLoopBatch(w, h, K, ptrs(0), [depth0] * S, kf_every=10, images_on_device=True, device=0)
            loop.set_keyframe_phases([((s % 4) * 10) // 4 for s in range(S)])
            if overlap:
                loop.set_pipeline(True)
            t = 1
            loop.run([ptrs(t + k) for k in range(10)]); t += 10
            capi.check(L.lsdhip_ctx_synchronize(loop.ctx_handle()))
            dts = []
            for _ in range(3):
                b = [ptrs(t + k) for k in range(30)]
                t0 = time.perf_counter(); loop.run(b); capi.check(L.lsdhip_ctx_synchronize(loop.ctx_handle())); dts.append(time.perf_counter() - t0); t += 30
            st = loop.stats(); loop.close()
            print("S=%d pipelined=%d: %.0f frames/s (ms per step %s), tracked good %d / %d" % (S, overlap, S * 30 / np.median(dts), " ".join("%.3f" % (x / 30 * 1e3) for x in dts), sum(q["tracked_good"] for q in st), sum(q["frames"] for q in st)), flush=True)
PY
