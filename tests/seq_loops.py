"""Sequence-level harness (SURVEY.md Appendix C level 5): the same single-sequence loop — the doSlam=false,
blockUntilMapped=true slice of SlamSystem (C/SlamSystem.cpp:890-1040 trackFrame, :739-828 doMappingIteration, :542-614
updateKeyframe, :458-490 createNewCurrentKeyframe) with a new keyframe every `kf_every` frames — driven once over the
oracle and once over the HIP classes, each side FEEDING ITS OWN OUTPUTS FORWARD (its own pose as the next initial
estimate, its own refPixelWasGood mask into its own depth update, its own propagated map as the next tracking
reference).  Test infrastructure only."""
import numpy as np

IDENT7 = np.array([1.0, 0, 0, 0, 0, 0, 0])
ODOMETRY_ITS = [5, 20, 50, 100, 0]


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class Sim3World:
    """keyframe -> world similarity (s, R, t): p_world = s R p_kf + t, chained through createKeyFrame's
    thisToParent = sim3FromSE3(oldToNew^-1, rescaleFactor) (C/DepthEstimation/DepthMap.cpp:1305)."""

    def __init__(self):
        self.s, self.R, self.t = 1.0, np.eye(3), np.zeros(3)

    def centre(self, frameToKF7):
        return self.s * (self.R @ np.asarray(frameToKF7[4:7])) + self.t

    def push_keyframe(self, newKFToOldKF7, rescale):
        Rk = quat_to_rot(newKFToOldKF7[:4])
        tk = np.asarray(newKFToOldKF7[4:7])
        self.t = self.s * (self.R @ tk) + self.t
        self.R = self.R @ Rk
        self.s = self.s * rescale


class Record:
    def __init__(self):
        self.frameToKF, self.centres, self.diverged, self.good, self.usage = [], [], [], [], []
        self.residual = []
        self.kf_frames, self.rescale, self.semidense = [], [], []      # per keyframe change
        self.final_semidense = None
        self.final_valid = None

    def trajectory(self):
        return np.array(self.centres)


def run_oracle(po, frames, depth0, K, n, kf_every=10, mode=None, params=None, L=None, live_queue=1):
    h, w = frames[0].shape
    mode = po.SSE if mode is None else mode
    kf = po.Frame(0, frames[0], K, L=L)
    kf.set_depth_gt(depth0)
    dm = po.DepthMap(w, h, K, params=params, L=L)
    dm.init_gt(kf)
    ref = po.TrackingReference(L=L)
    ref.import_frame(kf)
    tr = po.SE3Tracker(w, h, K, params=params, mode=mode, L=L)
    tr.set_max_its(ODOMETRY_ITS)
    rec, world = Record(), Sim3World()
    last, since = IDENT7.copy(), 0
    queue = []          # live_queue > 1: the mapper finds several tracked frames waiting (C/SlamSystem.cpp:559-571) — as SlamLoop models it
    for i in range(1, n + 1):
        f = po.Frame(i, frames[i % len(frames)], K, L=L)
        ref.import_frame(kf)                       # the keyframe's depth changed in the last mapping iteration
        r = tr.track(ref, f, last)
        est = np.array(r.frameToRef)
        rec.frameToKF.append(est)
        rec.centres.append(world.centre(est))
        rec.diverged.append(bool(r.diverged))
        rec.good.append(bool(r.trackingWasGood))
        rec.usage.append(r.pointUsage)
        rec.residual.append(r.lastResidual)
        if r.diverged:
            break
        since += 1
        if since >= kf_every:
            dm.finalize()
            s = dm.create_keyframe(f)
            world.push_keyframe(est, s)
            rec.kf_frames.append(i)
            rec.rescale.append(s)
            rec.semidense.append(int((dm.get()["isValid"] > 0).sum()))
            kf = f
            last, since = IDENT7.copy(), 0
            queue = []
        else:
            queue = (queue + [f])[-max(1, live_queue):]
            dm.update(list(queue))
            f.clear_wasgood()
            last = est
    m = dm.get()
    rec.final_valid = m["isValid"] > 0
    rec.final_semidense = int(rec.final_valid.sum())
    rec.final_map = m
    return rec


def run_hip(la, ctx, frames, depth0, n, kf_every=10, live_queue=1):
    kf = la.Frame(ctx, 0, frames[0])
    kf.setDepthFromGroundTruth(depth0)
    dm = la.DepthMap(ctx)
    dm.initializeFromGTDepth(kf)
    ref = la.TrackingReference()
    ref.importFrame(kf)
    tr = la.SE3Tracker(ctx)
    tr.set_maxItsPerLvl(ODOMETRY_ITS)
    rec, world = Record(), Sim3World()
    last, since = IDENT7.copy(), 0
    queue = []
    for i in range(1, n + 1):
        f = la.Frame(ctx, i, frames[i % len(frames)])
        ref.importFrame(kf)
        est = np.array(tr.trackFrame(ref, f, last))
        rec.frameToKF.append(est)
        rec.centres.append(world.centre(est))
        rec.diverged.append(bool(tr.diverged))
        rec.good.append(bool(tr.trackingWasGood))
        rec.usage.append(tr.pointUsage)
        rec.residual.append(tr.lastResidual)
        if tr.diverged:
            break
        since += 1
        if since >= kf_every:
            dm.finalizeKeyFrame()
            s = dm.createKeyFrame(f)
            world.push_keyframe(est, s)
            rec.kf_frames.append(i)
            rec.rescale.append(s)
            rec.semidense.append(int((dm.currentDepthMap()["isValid"] > 0).sum()))
            kf = f
            last, since = IDENT7.copy(), 0
            queue = []
        else:
            queue = (queue + [f])[-max(1, live_queue):]
            dm.updateKeyframe(list(queue))
            f.clear_refPixelWasGood()
            last = est
    m = dm.currentDepthMap()
    rec.final_valid = m["isValid"] > 0
    rec.final_semidense = int(rec.final_valid.sum())
    rec.final_map = m
    return rec


def rmse(a, b):
    a, b = np.asarray(a), np.asarray(b)
    n = min(len(a), len(b))
    return float(np.sqrt(np.mean(np.sum((a[:n] - b[:n]) ** 2, axis=1))))
