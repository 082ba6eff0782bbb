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
        self.evals = []                # numEvaluations of every trackFrame call (the LM loop's control flow, in one number)
        self.kf_frames, self.rescale, self.semidense = [], [], []      # per keyframe change
        self.final_semidense = None
        self.final_valid = None

    def trajectory(self):
        return np.array(self.centres)


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def relative_pose(kfToParent7, kf_scale, frameToParent7):
    """se3FromSim3(newKeyframe.camToWorld^-1 * frame.camToWorld) for two frames tracked on the same parent (C/SlamSystem.cpp:918-920):
    the new keyframe's thisToParent carries the createKeyFrame rescale factor as its Sim3 scale, the frame's has scale 1."""
    qk, tk = np.asarray(kfToParent7[:4]), np.asarray(kfToParent7[4:7])
    qf, tf = np.asarray(frameToParent7[:4]), np.asarray(frameToParent7[4:7])
    qi = qk * np.array([1.0, -1, -1, -1])
    q = quat_mul(qi, qf)
    q = q / np.linalg.norm(q)
    t = quat_to_rot(qi) @ (tf - tk) / kf_scale
    return np.concatenate([q, t])


class OracleSide:
    """the loop's operations on the oracle (oracle/pyoracle.py)"""

    def __init__(self, po, frames, depth0, K, mode=None, params=None, L=None):
        self.po, self.frames, self.K, self.L, self.params = po, frames, K, L, params
        h, w = frames[0].shape
        self.kf0 = po.Frame(0, frames[0], K, L=L)
        self.kf0.set_depth_gt(depth0)
        self.dm = po.DepthMap(w, h, K, params=params, L=L)
        self.dm.init_gt(self.kf0)
        self.ref = po.TrackingReference(L=L)
        self.tr = po.SE3Tracker(w, h, K, params=params, mode=po.SSE if mode is None else mode, L=L)
        self.tr.set_max_its(ODOMETRY_ITS)

    def frame(self, i):
        return self.po.Frame(i, self.frames[i % len(self.frames)], self.K, L=self.L)

    def import_ref(self, kf, snapshot):
        self.ref.import_frame(kf)
        if snapshot:
            # the reference's TrackingReference materialises its point clouds at the first trackFrame after importFrame
            # (makePointCloud, C/Tracking/TrackingReference.cpp:96-147): taken now, before the mapper touches the keyframe again
            for lvl in (4, 3, 2, 1):
                self.ref.pointcloud(lvl)

    def track(self, f, init):
        r = self.tr.track(self.ref, f, init)
        return np.array(r.frameToRef), bool(r.diverged), bool(r.trackingWasGood), r.pointUsage, r.lastResidual, int(r.numEvaluations)

    def depth_flag(self, kf):
        return kf.stats()["depthHasBeenUpdatedFlag"] != 0

    def clear_depth_flag(self, kf):
        st = kf.stats()
        kf.set_counters(int(st["numFramesTrackedOnThis"]), int(st["numMappedOnThis"]), int(st["numMappedOnThisTotal"]), 0)

    def finalize(self):
        self.dm.finalize()

    def create_keyframe(self, f):
        return self.dm.create_keyframe(f)

    def update(self, fs):
        self.dm.update(list(fs))

    def clear_wasgood(self, f):
        f.clear_wasgood()

    def get_map(self):
        return self.dm.get()


class HipSide:
    """the same operations on the HIP classes (lsd_slam_amd/slam.py)"""

    def __init__(self, la, ctx, frames, depth0):
        self.la, self.ctx, self.frames = la, ctx, frames
        self.kf0 = la.Frame(ctx, 0, frames[0])
        self.kf0.setDepthFromGroundTruth(depth0)
        self.dm = la.DepthMap(ctx)
        self.dm.initializeFromGTDepth(self.kf0)
        self.ref = la.TrackingReference()
        self.tr = la.SE3Tracker(ctx)
        self.tr.set_maxItsPerLvl(ODOMETRY_ITS)

    def frame(self, i):
        return self.la.Frame(self.ctx, i, self.frames[i % len(self.frames)])

    def import_ref(self, kf, snapshot):
        self.ref.importFrame(kf)          # pipelined contexts: lsdhip_frame_publish_depth — the snapshot moment

    def track(self, f, init):
        est = np.array(self.tr.trackFrame(self.ref, f, init))
        return est, bool(self.tr.diverged), bool(self.tr.trackingWasGood), self.tr.pointUsage, self.tr.lastResidual, int(self.tr.last.numEvaluations)

    def depth_flag(self, kf):
        return kf.depthHasBeenUpdatedFlag()

    def clear_depth_flag(self, kf):
        kf.clearDepthHasBeenUpdatedFlag()

    def finalize(self):
        self.dm.finalizeKeyFrame()

    def create_keyframe(self, f):
        return self.dm.createKeyFrame(f)

    def update(self, fs):
        self.dm.updateKeyframe(list(fs))

    def clear_wasgood(self, f):
        f.clear_refPixelWasGood()

    def get_map(self):
        return self.dm.currentDepthMap()


def run_loop(side, n, kf_every=10, live_queue=1, lag=0, clear_flag=False, init_perturb=None):
    """lag = 0: blockUntilMapped — track frame t, then its mapping iteration; frame t + 1 sees what map(t) left.
    lag = 1: the mapper one frame behind the tracker (lsd_slam_hip::SlamLoop pipelined; C/SlamSystem.cpp:907-920, :559-566 with
    blockUntilMapped == false): frame t + 1 is tracked against the keyframe / depth version map(t - 1) left, map(t) runs "beside" it;
    the frame after a keyframe change is tracked on the old keyframe and dropped by the mapper; the first frame on the new keyframe
    starts from se3FromSim3(newKF^-1 * lastTrackedFrame).
    clear_flag: SlamSystem::trackFrame's import-and-clear of depthHasBeenUpdatedFlag (:907-912), so that EVERY updateKeyframe runs
    Frame::setDepth (DepthMap.cpp:1148-1154) — implied by lag = 1; the historic lag-0 cases leave the flag alone (the tracker then
    sees new depth at keyframe changes only)."""
    clear_flag = clear_flag or lag > 0
    rec = Record()
    rec.tracked_on, rec.dropped = [], []
    map_kf = track_kf = side.kf0
    side.import_ref(track_kf, lag > 0)
    if clear_flag:
        side.clear_depth_flag(track_kf)
    world_track = Sim3World()            # world transform of the keyframe the tracker uses / the mapper holds
    world_map = world_track
    pending = None                       # lag 1: (new keyframe, its frameToOldKF, rescale factor, its world transform)
    last, since = IDENT7.copy(), 0
    if init_perturb is not None:
        last[4:7] += np.asarray(init_perturb)      # sensitivity probe: the first frame's initial estimate moved by a few float32 ulps
    queue = []          # live_queue > 1: the mapper finds several tracked frames waiting (C/SlamSystem.cpp:559-571) — as SlamLoop models it
    for i in range(1, n + 1):
        f = side.frame(i)
        if lag == 0:
            if clear_flag:
                if side.depth_flag(track_kf):
                    side.import_ref(track_kf, False)
                    side.clear_depth_flag(track_kf)
            else:
                side.import_ref(track_kf, False)           # the keyframe's depth changed in the last mapping iteration
        tracked_on = track_kf
        est, diverged, good, usage, residual, evals = side.track(f, last)
        rec.frameToKF.append(est)
        rec.centres.append(world_track.centre(est))
        rec.diverged.append(diverged)
        rec.good.append(good)
        rec.usage.append(usage)
        rec.residual.append(residual)
        rec.evals.append(evals)
        rec.tracked_on.append(rec.kf_frames[-1] if (rec.kf_frames and tracked_on is map_kf) else (0 if tracked_on is side.kf0 else -1))
        if diverged:
            break
        if lag > 0:
            # what the tracking thread finds when it comes back: the mapping iteration queued one frame ago has finished
            if pending is not None:
                kf_new, kf_est, kf_scale, kf_world = pending
                last = relative_pose(kf_est, kf_scale, est)
                track_kf, world_track, pending = kf_new, kf_world, None
                side.import_ref(track_kf, True)
                side.clear_depth_flag(track_kf)
            else:
                last = est
                if side.depth_flag(track_kf):
                    side.import_ref(track_kf, True)
                    side.clear_depth_flag(track_kf)
        since += 1
        if tracked_on is not map_kf:
            side.clear_wasgood(f)        # tracked on the keyframe the mapper has just replaced: popped unmapped (:559-566)
            rec.dropped.append(i)
            continue
        if since >= kf_every:
            side.finalize()
            s = side.create_keyframe(f)
            new_world = Sim3World()
            new_world.s, new_world.R, new_world.t = world_map.s, world_map.R.copy(), world_map.t.copy()
            new_world.push_keyframe(est, s)
            rec.kf_frames.append(i)
            rec.rescale.append(s)
            rec.semidense.append(int((side.get_map()["isValid"] > 0).sum()))
            map_kf, world_map = f, new_world
            since = 0
            queue = []
            if lag > 0:
                pending = (f, est, s, new_world)
            else:
                track_kf, world_track = f, new_world
                last = IDENT7.copy()
                if clear_flag:
                    side.import_ref(track_kf, False)
                    side.clear_depth_flag(track_kf)
        else:
            queue = (queue + [f])[-max(1, live_queue):]
            side.update(queue)
            side.clear_wasgood(f)
            if lag == 0:
                last = est
    m = side.get_map()
    rec.final_valid = m["isValid"] > 0
    rec.final_semidense = int(rec.final_valid.sum())
    rec.final_map = m
    return rec


def run_oracle(po, frames, depth0, K, n, kf_every=10, mode=None, params=None, L=None, live_queue=1, lag=0, clear_flag=False, init_perturb=None):
    return run_loop(OracleSide(po, frames, depth0, K, mode=mode, params=params, L=L), n, kf_every, live_queue, lag, clear_flag, init_perturb)


def run_hip(la, ctx, frames, depth0, n, kf_every=10, live_queue=1, lag=0, clear_flag=False):
    """lag = 1 expects a pipelined, asynchronous context (ctx.set_pipeline(True); ctx.set_async(True)): then the mapping calls of
    frame t are merely queued when frame t + 1's tracking starts, and the loop's results must not depend on that"""
    return run_loop(HipSide(la, ctx, frames, depth0), n, kf_every, live_queue, lag, clear_flag)


def rmse(a, b):
    a, b = np.asarray(a), np.asarray(b)
    n = min(len(a), len(b))
    return float(np.sqrt(np.mean(np.sum((a[:n] - b[:n]) ** 2, axis=1))))


def replay_centres(frameToKF, rescales, kf_every=10, lag=0):
    """Camera centres in the world frame from a loop's per-frame poses and its createKeyFrame rescale factors alone — the bookkeeping
    run_loop does while it runs (Sim3World chained at every keyframe change; lag = 1: the frame that follows a keyframe change is still
    tracked on the old keyframe).  For loops that only report poses and keyframe logs (lsd_slam_hip::SlamLoopBatch through the C driver)."""
    world_track = Sim3World()
    world_map = world_track
    pending = None
    out, since, k = [], 0, 0
    for i, est in enumerate(frameToKF, start=1):
        tracked_on_map = world_track is world_map
        out.append(world_track.centre(est))
        if pending is not None:
            world_track, pending = pending, None
        since += 1
        if not tracked_on_map:
            continue
        if since >= kf_every:
            nw = Sim3World()
            nw.s, nw.R, nw.t = world_map.s, world_map.R.copy(), world_map.t.copy()
            nw.push_keyframe(est, rescales[k])
            k += 1
            world_map, since = nw, 0
            if lag > 0:
                pending = nw
            else:
                world_track = nw
    return np.array(out)
