#!/usr/bin/env python
"""Developer probe (round 6): trackFrameSim3 on the device against the oracle's three arithmetic modes (SSE with IEEE reciprocals, SSE with
_mm_rcp_ps, scalar) — whole calls and PREFIXES of the LM loop (the first n iterations of the coarsest level, then each finer level added)
— to see where the two LM loops part and what the reference's own spread is.  python tools/sim3_parity_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lsd_slam_amd as hip
import oracle.pyoracle as po
from test_sim3_gpu import pair
w, h = 320, 240


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def pd(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a[4:7] - b[4:7]), min(np.linalg.norm(a[:4] - b[:4]), np.linalg.norm(a[:4] + b[:4])), rel(a[7], b[7])


for scale in (1.0, 1.25, 0.8):
    P = pair(po, hip, w, h, 3, scale)
    init = P["exp"].copy(); init[7] = 1.0
    tg = hip.Sim3Tracker(P["ctx"])
    schedules = [("L3 x%d" % n, [0, 0, 0, n, 0], 3, 3) for n in (1, 2, 3, 4, 6, 8, 12, 100)] + [("L3-2", [5, 20, 50, 100, 100], 3, 2), ("L3-1 (full)", [5, 20, 50, 100, 100], 3, 1)]
    for name, its, s0, s1 in schedules:
        out = {}
        for mname, mode in (("exact", po.SSE_EXACT_RCP), ("sse", po.SSE), ("scalar", po.SCALAR)):
            to = po.Sim3Tracker(w, h, P["K"], mode=mode); to.set_max_its(its)
            out[mname] = to.track(P["ra"], P["fb"], init, s0, s1)
        tg.setMaxItsPerLvl(its)
        got, rg = tg.trackFrameSim3(P["ga"], P["gb"], init, s0, s1)
        ro = out["exact"]
        want = np.array(ro.frameToRef)
        dt, dq, ds = pd(got, want)
        sp = [pd(np.array(out[m].frameToRef), want) for m in ("sse", "scalar")]
        print("scale %.2f %-12s evals hip %3d exact %3d sse %3d scalar %3d | hip-exact dt %.1e dq %.1e ds %.1e | sse-exact dt %.1e ds %.1e | scalar-exact dt %.1e ds %.1e | res rel hip %.1e sse %.1e sc %.1e | resD hip %.1e sse %.1e sc %.1e | H rel hip %.1e sc %.1e" % (
            scale, name, rg.numEvaluations, ro.numEvaluations, out["sse"].numEvaluations, out["scalar"].numEvaluations, dt, dq, ds,
            sp[0][0], sp[0][2], sp[1][0], sp[1][2],
            rel(rg.lastResidual, ro.lastResidual), rel(out["sse"].lastResidual, ro.lastResidual), rel(out["scalar"].lastResidual, ro.lastResidual),
            rel(rg.lastDepthResidual, ro.lastDepthResidual), rel(out["sse"].lastDepthResidual, ro.lastDepthResidual), rel(out["scalar"].lastDepthResidual, ro.lastDepthResidual),
            np.abs(np.array(rg.lastSim3Hessian) - np.array(ro.hessian)).max() / np.abs(np.array(ro.hessian)).max(),
            np.abs(np.array(out["scalar"].hessian) - np.array(ro.hessian)).max() / np.abs(np.array(ro.hessian)).max()))
