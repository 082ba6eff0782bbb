#!/usr/bin/env python
"""Developer check (round 6): per-frame lastResidual / pose of the S-sequence C++ loop against the oracle loop and the Python-driven HIP loop."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lsd_slam_amd as la
from lsd_slam_amd import synth
from lsd_slam_amd.driver import DriverLoopBatch
import oracle.pyoracle as po
import seq_loops as sl
w, h, n, S = 640, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 6, 8
pipelined = len(sys.argv) > 2 and sys.argv[2] == "1"
seqs = [synth.make_sequence(w, h, 50, s, "S1") for s in range(S)]
K = seqs[0][2]
imgs = [[np.ascontiguousarray(f) for f in q[0]] for q in seqs]
bl = DriverLoopBatch(w, h, K, [imgs[s][0].ctypes.data for s in range(S)], [seqs[s][1] for s in range(S)], kf_every=10, images_on_device=False)
bl.set_pipeline(pipelined)
res = [[] for _ in range(S)]
for t in range(1, n + 1):
    done, poses = bl.run([[imgs[s][t % 50].ctypes.data for s in range(S)]], want_poses=True)
    for s in range(S):
        r = bl.last_result(s)
        res[s].append((r.lastResidual, r.pointUsage, r.numEvaluations, poses[0, s].copy()))
bl.close()
for s in range(S):
    o = sl.run_oracle(po, seqs[s][0], seqs[s][1], K, n, mode=po.SSE, lag=1 if pipelined else 0, clear_flag=True)
    ctx = la.Context(w, h, K)
    if pipelined:
        ctx.set_pipeline(True); ctx.set_async(True)
    g = sl.run_hip(la, ctx, seqs[s][0], seqs[s][1], n, lag=1 if pipelined else 0, clear_flag=True)
    ctx.set_pipeline(False) if pipelined else None
    print("seq %d" % s)
    for t in range(n):
        print("  frame %2d: lastResidual batch %.6f  python-hip %.6f  oracle %.6f | usage %.5f %.5f %.5f | |t| diff batch-oracle %.1e hip-oracle %.1e" % (
            t + 1, res[s][t][0], g.residual[t], o.residual[t], res[s][t][1], g.usage[t], o.usage[t],
            np.abs(res[s][t][3][4:7] - o.frameToKF[t][4:7]).max(), np.abs(g.frameToKF[t][4:7] - o.frameToKF[t][4:7]).max()))
