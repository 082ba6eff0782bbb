#!/usr/bin/env python
"""Developer tool: where does the pipelined loop first go wrong when a mapping iteration really runs beside the next tracking job?
A lag-1 loop without keyframe changes is driven from Python on a pipelined context; every frame's pose, refPixelWasGood mask, the
hypothesis map after its mapping iteration and the keyframe planes Frame::setDepth wrote are recorded (everything is drained before
the read-backs, between the tracking job and the next mapping iteration).  Run twice and compare:
    LSDHIP_PIPE_GATE=1 python tools/pipe_overlap_debug.py run /tmp/a.npz     # forced overlap (the mapping kernels wait for the next tracking job's start)
    python tools/pipe_overlap_debug.py run /tmp/b.npz                        # the Python loop's own pace: every mapping iteration has finished before the next job
    python tools/pipe_overlap_debug.py cmp /tmp/a.npz /tmp/b.npz
(These switches live in the developer build only: python -c "from lsd_slam_amd import build; build.build_variant('devtools', ['LSD_DEVTOOLS'])", then LSDHIP_LIB=lsd_slam_amd/liblsdhip_devtools.so LD_PRELOAD=$LSDHIP_LIB.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(path, n=24, light=False, device_images=False):
    import lsd_slam_amd as la
    from common import sequence, ODOMETRY_ITS
    w, h = 640, 480
    frames, depth0, K, gt = sequence(w, h, 41)
    ctx = la.Context(w, h, K)
    ctx.set_pipeline(True)
    ctx.set_async(True)
    kf = la.Frame(ctx, 0, frames[0])
    kf.setDepthFromGroundTruth(depth0)
    dm = la.DepthMap(ctx)
    dm.initializeFromGTDepth(kf)
    ref = la.TrackingReference()
    ref.importFrame(kf)
    kf.clearDepthHasBeenUpdatedFlag()
    tr = la.SE3Tracker(ctx)
    tr.set_maxItsPerLvl(ODOMETRY_ITS)
    last = la.IDENTITY.copy()
    out = {}
    if device_images:
        import torch
        dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    mk = lambda i: la.Frame(ctx, i, device_ptr=dev[i % dev.shape[0]].data_ptr()) if device_images else la.Frame(ctx, i, frames[i % len(frames)])
    nxt = mk(1)
    prev = None
    for i in range(1, n + 1):
        f = nxt
        est = np.array(tr.trackFrame(ref, f, last))
        out["pose%02d" % i] = est
        out["evals%02d" % i] = np.array([tr.last.numEvaluations, tr.last.numWarpUpdates])
        if not light:
            ctx.synchronize()                       # tracking job i and mapping iteration i - 1 are complete
            out["mask%02d" % i] = f.refPixelWasGoodNoCreate()
        if prev is not None and not light:
            m = dm.currentDepthMap()
            for k in ("isValid", "validity_counter", "blacklisted", "idepth", "idepth_var", "idepth_smoothed", "idepth_var_smoothed"):
                out["map%02d_%s" % (i - 1, k)] = m[k].copy()
            for lvl in (0, 1, 2, 3, 4):
                out["kfid%02d_L%d" % (i - 1, lvl)] = kf.idepth(lvl)
                out["kfvar%02d_L%d" % (i - 1, lvl)] = kf.idepthVar(lvl)
        if kf.depthHasBeenUpdatedFlag():
            ref.importFrame(kf)
            kf.clearDepthHasBeenUpdatedFlag()
        last = est
        nxt = mk(i + 1)    # before the mapping iteration is queued: the mapping stream is idle
        if not light:
            for lvl in (1, 4):
                out["grad%02d_L%d" % (i + 1, lvl)] = nxt.gradients(lvl)
        dm.updateKeyframe([f])                  # LSDHIP_PIPE_GATE: held until the next tracking job starts
        f.clear_refPixelWasGood()
        prev = f
    tr.trackFrame(ref, nxt, last)     # releases the gate the last mapping iteration waits at
    ctx.synchronize()
    np.savez_compressed(path, **out)
    print("saved", path, len(out), "arrays")


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    keys = sorted(A.files, key=lambda k: (int("".join(ch for ch in k.split("_")[0] if ch.isdigit())), k))
    bad = 0
    for k in keys:
        x, y = A[k], B[k]
        if x.shape != y.shape or not np.array_equal(x.view(np.uint8) if x.dtype != bool else x, y.view(np.uint8) if y.dtype != bool else y):
            nd = int((x != y).sum()) if x.shape == y.shape else -1
            where = np.argwhere(x != y)[:3].tolist() if x.shape == y.shape else []
            print("DIFF %-28s %8d of %8d elements differ, first at %s" % (k, nd, x.size, where))
            bad += 1
            if bad > 25:
                break
    print("compared %d arrays, %d differ" % (len(keys), bad))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 24, light="light" in sys.argv, device_images="dev" in sys.argv)
    else:
        cmp(sys.argv[2], sys.argv[3])
