"""Host-side mirror of the reference's class interface for the hot path, on top of the C ABI (include/lsdhip.h).

Same names, argument meaning and error behaviour as the reference so that parity tests read like the reference's
call sites (`C/` = lsd_slam_core/src/):

    Frame(id, image, ctx)                          C/DataStructures/Frame.h:43
    TrackingReference.importFrame(kf)              C/Tracking/TrackingReference.h:49
    SE3Tracker.trackFrame(reference, frame, init)  C/Tracking/SE3Tracker.h:65-68
    DepthMap.updateKeyframe / createKeyFrame / finalizeKeyFrame / initializeFromGTDepth …
                                                   C/DepthEstimation/DepthMap.h:47-98

Poses are numpy double[7] = (qw,qx,qy,qz,tx,ty,tz).  All per-pixel work happens in liblsdhip.so on the GPU; this
file only marshals arguments (no numerical fallback of any kind).
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import HYP_DTYPE, check

IDENTITY = np.array([1.0, 0, 0, 0, 0, 0, 0])


class Context:
    """(w, h, K) + device + stream shared by Frame / SE3Tracker / DepthMap objects."""

    def __init__(self, w, h, K, device=0, params=None):
        self.L = capi.lib()
        self.w, self.h = int(w), int(h)
        self.K = np.ascontiguousarray(K, dtype=np.float32)
        self.params = capi.Params()
        self.L.lsdhip_default_params(C.byref(self.params))
        if params:
            for k, v in params.items():
                setattr(self.params, k, v)
        h_ = C.c_void_p()
        check(self.L.lsdhip_ctx_create(device, self.w, self.h, self.K.ctypes.data, C.byref(self.params), C.byref(h_)),
              allow_positive=False)
        self.h_ = h_

    def close(self):
        if getattr(self, "h_", None):
            self.L.lsdhip_ctx_destroy(self.h_)
            self.h_ = None

    def synchronize(self):
        check(self.L.lsdhip_ctx_synchronize(self.h_))

    def set_async(self, on=True):
        """DepthMap calls return once queued (lsdhip_ctx_set_async)"""
        check(self.L.lsdhip_ctx_set_async(self.h_, int(bool(on))))

    def set_pipeline(self, on=True):
        """tracking stream beside mapping stream (lsdhip_ctx_set_pipeline): the reference's two threads with blockUntilMapped == false"""
        check(self.L.lsdhip_ctx_set_pipeline(self.h_, int(bool(on))))

    def lanes_begin(self, n):
        """open a lane region: independent call chains (per-sequence keyframe changes) go to n side streams (lsdhip_ctx_lanes_begin)"""
        check(self.L.lsdhip_ctx_lanes_begin(self.h_, int(n)))

    def lane_select(self, lane):
        check(self.L.lsdhip_ctx_lane_select(self.h_, int(lane)))

    def lanes_end(self):
        check(self.L.lsdhip_ctx_lanes_end(self.h_))

    def reserve_frames(self, n):
        """frame-memory pool: n arenas allocated ahead (lsdhip_ctx_reserve_frames)"""
        check(self.L.lsdhip_ctx_reserve_frames(self.h_, int(n)), False)

    def intrinsics(self, level):
        out = np.zeros(8, np.float32)
        check(self.L.lsdhip_ctx_intrinsics(self.h_, level, out.ctypes.data))
        return out

    def stream(self):
        return self.L.lsdhip_ctx_stream(self.h_)

    def prof_enable(self, on=True):
        check(self.L.lsdhip_prof_enable(self.h_, int(on)))

    def prof_reset(self):
        check(self.L.lsdhip_prof_reset(self.h_))

    def prof_read(self):
        ms, n, b = C.c_double(), C.c_longlong(), C.c_double()
        check(self.L.lsdhip_prof_read(self.h_, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value


class Frame:
    """Device-resident frame: pyramids are built on the GPU at construction."""

    def __init__(self, ctx, id_, image=None, device_ptr=None):
        self.ctx = ctx
        self.L = ctx.L
        h_ = C.c_void_p()
        if device_ptr is not None:
            check(self.L.lsdhip_frame_create_from_device(ctx.h_, id_, C.c_void_p(device_ptr), C.byref(h_)), False)
        else:
            img = np.ascontiguousarray(image, dtype=np.uint8)
            if img.shape != (ctx.h, ctx.w):
                raise ValueError("image must be %dx%d uint8" % (ctx.w, ctx.h))
            check(self.L.lsdhip_frame_create(ctx.h_, id_, img.ctypes.data, C.byref(h_)), False)
        self.h_ = h_
        self._id = id_
        self._parent = None

    @classmethod
    def createBatch(cls, ctx, ids, images=None, device_ptrs=None):
        """the new frames of several sequences in two launches (lsdhip_frame_create_batch) -> list of Frame"""
        n = len(ids)
        ida = (C.c_int * n)(*[int(v) for v in ids])
        out = (C.c_void_p * n)()
        if device_ptrs is not None:
            ptrs = (C.c_void_p * n)(*[int(p) for p in device_ptrs])
            check(ctx.L.lsdhip_frame_create_batch(ctx.h_, n, ida, ptrs, 1, out), False)
        else:
            imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
            for im in imgs:
                if im.shape != (ctx.h, ctx.w):
                    raise ValueError("image must be %dx%d uint8" % (ctx.w, ctx.h))
            ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
            check(ctx.L.lsdhip_frame_create_batch(ctx.h_, n, ida, ptrs, 0, out), False)
        frames = []
        for j in range(n):
            f = cls.__new__(cls)
            f.ctx, f.L, f.h_, f._id, f._parent = ctx, ctx.L, C.c_void_p(out[j]), int(ids[j]), None
            frames.append(f)
        return frames

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "h_", None) and getattr(self.ctx, "h_", None):
            self.L.lsdhip_frame_destroy(self.h_)
        self.h_ = None

    def id(self):
        return self._id

    def width(self, level=0):
        return self.ctx.w >> level

    def height(self, level=0):
        return self.ctx.h >> level

    def _plane(self, what, level):
        shape = (self.height(level), self.width(level), 4) if what == 1 else (self.height(level), self.width(level))
        out = np.zeros(shape, np.float32)
        check(self.L.lsdhip_frame_download(self.h_, what, level, out.ctypes.data), False)
        return out

    def image(self, level=0):
        return self._plane(0, level)

    def gradients(self, level=0):
        return self._plane(1, level)

    def maxGradients(self, level=0):
        return self._plane(2, level)

    def idepth(self, level=0):
        return self._plane(3, level)

    def idepthVar(self, level=0):
        return self._plane(4, level)

    def referenceBlocks(self, level):
        """(offsets uint8 [blocks, 256], counts int32 [blocks]) of the level's reference blocks (lsdhip_frame_download what = 5): per 256
        consecutive pixels the in-block offsets of the valid reference pixels, compacted in pixel order (returned in list order: on the device
        slot s of a block sits in byte (s mod 64) * 4 + s // 64, so that a lane's 4-byte word holds slots l, l + 64, l + 128, l + 192)."""
        nblk = (self.width(level) * self.height(level) + 255) // 256
        raw = np.zeros(nblk * 260, np.uint8)
        check(self.L.lsdhip_frame_download(self.h_, 5, level, raw.ctypes.data), False)
        offs = raw[:nblk * 256].reshape(nblk, 64, 4).transpose(0, 2, 1).reshape(nblk, 256)
        return offs.copy(), raw[nblk * 256:].view(np.int32).copy()

    def gradientCandidates(self):
        """(offsets uint16 [groups, 1024], counts uint16 [groups]) of the frame's gradient candidates (lsdhip_frame_download what = 6)."""
        ng = (self.width(0) * self.height(0) + 1023) // 1024
        raw = np.zeros(ng * 1025, np.uint16)
        check(self.L.lsdhip_frame_download(self.h_, 6, 0, raw.ctypes.data), False)
        return raw[:ng * 1024].reshape(ng, 1024).copy(), raw[ng * 1024:].copy()

    def setDepthFromGroundTruth(self, depth, cov_scale=1.0):
        d = np.ascontiguousarray(depth, dtype=np.float32)
        check(self.L.lsdhip_frame_set_depth_gt(self.h_, d.ctypes.data, cov_scale), False)

    def setDepthPlanes(self, idepth, idepthVar):
        a = np.ascontiguousarray(idepth, dtype=np.float32)
        b = np.ascontiguousarray(idepthVar, dtype=np.float32)
        check(self.L.lsdhip_frame_set_depth_planes(self.h_, a.ctypes.data, b.ctypes.data), False)

    def setMaxGradients(self, plane):
        p = np.ascontiguousarray(plane, dtype=np.float32)
        check(self.L.lsdhip_frame_set_maxgrad(self.h_, p.ctypes.data), False)

    def refPixelWasGoodNoCreate(self):
        out = np.zeros((self.height(1), self.width(1)), np.uint8)
        rc = check(self.L.lsdhip_frame_get_wasgood(self.h_, out.ctypes.data))
        return out if rc == 1 else None

    def set_refPixelWasGood(self, m):
        m = np.ascontiguousarray(m, dtype=np.uint8)
        check(self.L.lsdhip_frame_set_wasgood(self.h_, m.ctypes.data), False)

    def clear_refPixelWasGood(self):
        check(self.L.lsdhip_frame_clear_wasgood(self.h_))

    def setPose(self, thisToParent_sim3, trackingParent, initialTrackedResidual=0.0):
        s = np.ascontiguousarray(thisToParent_sim3, dtype=np.float64)
        self._parent = trackingParent
        check(self.L.lsdhip_frame_set_pose(self.h_, s.ctypes.data, trackingParent.h_ if trackingParent else None,
                                           initialTrackedResidual), False)

    def thisToParent_raw(self):
        out = np.zeros(8, np.float64)
        check(self.L.lsdhip_frame_get_pose(self.h_, out.ctypes.data))
        return out

    def relativePoseTo(self, reference):
        """se3FromSim3(reference.camToWorld^-1 * self.camToWorld) (C/SlamSystem.cpp:918-920) for a frame tracked on `reference` or on
        the same parent as `reference` (lsdhip_frame_relative_pose)"""
        out = np.zeros(7, np.float64)
        check(self.L.lsdhip_frame_relative_pose(reference.h_, self.h_, out.ctypes.data))
        return out

    def depthHasBeenUpdatedFlag(self):
        return check(self.L.lsdhip_frame_depth_updated(self.h_)) != 0

    def clearDepthHasBeenUpdatedFlag(self):
        check(self.L.lsdhip_frame_clear_depth_updated(self.h_))

    def stats(self):
        out = np.zeros(8, np.float32)
        check(self.L.lsdhip_frame_stats(self.h_, out.ctypes.data))
        keys = ["initialTrackedResidual", "meanIdepth", "numPoints", "numFramesTrackedOnThis", "numMappedOnThis",
                "numMappedOnThisTotal", "depthHasBeenUpdatedFlag", "reserved"]
        return dict(zip(keys, out.tolist()))

    def clearDepthHasBeenUpdatedFlag(self):
        """currentKeyFrame->depthHasBeenUpdatedFlag = false (C/SlamSystem.cpp:910, when the tracking reference is re-imported)"""
        s = self.stats()
        self.setCounters(int(s["numFramesTrackedOnThis"]), int(s["numMappedOnThis"]), int(s["numMappedOnThisTotal"]), False)

    def setCounters(self, numFramesTrackedOnThis, numMappedOnThis, numMappedOnThisTotal, depthHasBeenUpdatedFlag):
        check(self.L.lsdhip_frame_set_counters(self.h_, numFramesTrackedOnThis, numMappedOnThis, numMappedOnThisTotal,
                                               int(depthHasBeenUpdatedFlag)))


class TrackingReference:
    """C/Tracking/TrackingReference.h — on the device the point cloud is generated on the fly inside the residual
    kernel, so this object only remembers which keyframe it refers to; makePointCloud() exports the compacted arrays
    in the reference's order for callers that want them (permaref, Sim3 tracker)."""

    def __init__(self):
        self.keyframe = None

    def importFrame(self, kf):
        """TrackingReference.cpp:71-87; on a pipelined context the hand-over of the mapping side's newest Frame::setDepth result to
        the tracker (lsdhip_frame_publish_depth)"""
        self.keyframe = kf
        if kf is not None:
            check(kf.L.lsdhip_frame_publish_depth(kf.h_))

    def invalidate(self):
        self.keyframe = None

    def makePointCloud(self, level):
        kf = self.keyframe
        n_max = kf.width(level) * kf.height(level)
        pos = np.zeros((n_max, 3), np.float32)
        cv = np.zeros((n_max, 2), np.float32)
        gr = np.zeros((n_max, 2), np.float32)
        idx = np.zeros(n_max, np.int32)
        n = check(kf.L.lsdhip_ref_pointcloud(kf.h_, level, pos.ctypes.data, cv.ctypes.data, gr.ctypes.data, idx.ctypes.data))
        return pos[:n].copy(), cv[:n].copy(), gr[:n].copy(), idx[:n].copy()


class SE3Tracker:
    def __init__(self, ctx):
        self.ctx = ctx
        self.L = ctx.L
        h_ = C.c_void_p()
        check(self.L.lsdhip_tracker_create(ctx.h_, C.byref(h_)), False)
        self.h_ = h_
        self.last = None

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "h_", None) and getattr(self.ctx, "h_", None):
            self.L.lsdhip_tracker_destroy(self.h_)
        self.h_ = None

    def set_maxItsPerLvl(self, its):
        its = np.ascontiguousarray(its, dtype=np.int32)
        check(self.L.lsdhip_tracker_set_max_its(self.h_, its.ctypes.data))

    def settings(self):
        """DenseDepthTrackerSettings of this tracker (a copy; write back with set_settings)"""
        st = capi.TrackerSettings()
        check(self.L.lsdhip_tracker_get_settings(self.h_, C.byref(st)))
        return st

    def set_settings(self, st):
        check(self.L.lsdhip_tracker_set_settings(self.h_, C.byref(st)))


    def set_speculation(self, trials, finest_level_workgroups=0):
        """LM retries evaluated per launch (1 = one evaluation per launch); see include/lsdhip.h"""
        check(self.L.lsdhip_tracker_set_speculation(self.h_, int(trials), int(finest_level_workgroups)))

    def set_batch_coarse_min_jobs(self, min_jobs):
        """throughput-mode batches of at least `min_jobs` jobs walk their coarse levels in one workgroup per job (0: never); see include/lsdhip.h"""
        check(self.L.lsdhip_tracker_set_batch_coarse_min_jobs(self.h_, int(min_jobs)))

    def launch_stats(self):
        """(evaluating launches of the last job, trials per launch)"""
        out = np.zeros(2, np.int32)
        check(self.L.lsdhip_tracker_launch_stats(self.h_, out.ctypes.data))
        return int(out[0]), int(out[1])

    def summary_stats(self):
        """(jobs polled, records incomplete when `done` arrived, longest wait ns, stale words, lowest / highest stale word index)"""
        out = np.zeros(6, np.int64)
        check(self.L.lsdhip_tracker_summary_stats(self.h_, out.ctypes.data))
        return [int(v) for v in out]

    def step_stats(self):
        """(k_track_step launches of the last job, steps inside the cluster kernel, cluster-kernel launches, most trials per step)"""
        out = np.zeros(4, np.int32)
        check(self.L.lsdhip_tracker_step_stats(self.h_, out.ctypes.data))
        return tuple(int(v) for v in out)

    def exec_stats(self):
        """diagnostics: (cluster-kernel jobs, give-ups rerun on the chain, switched off, evaluations per level 0..4)"""
        out = np.zeros(8, np.int32)
        check(self.L.lsdhip_tracker_exec_stats(self.h_, out.ctypes.data))
        return int(out[0]), int(out[1]), bool(out[2]), out[3:8].tolist()

    def _publish(self, r):
        self.last = r
        self.pointUsage = r.pointUsage
        self.lastGoodCount = r.lastGoodCount
        self.lastBadCount = r.lastBadCount
        self.lastMeanRes = r.lastMeanRes
        self.lastResidual = r.lastResidual
        self.affineEstimation_a = r.affineEstimation_a
        self.affineEstimation_b = r.affineEstimation_b
        self.diverged = bool(r.diverged)
        self.trackingWasGood = bool(r.trackingWasGood)

    def trackFrame(self, reference, frame, frameToReference_initialEstimate):
        """Returns frameToReference (double[7]); identity when diverged, like the reference (SE3Tracker.cpp:324-329)."""
        init = np.ascontiguousarray(frameToReference_initialEstimate, dtype=np.float64)
        r = capi.TrackResult()
        check(self.L.lsdhip_tracker_track(self.h_, reference.keyframe.h_, frame.h_, init.ctypes.data, C.byref(r)))
        self._publish(r)
        if not r.diverged:
            frame._parent = reference.keyframe
        return np.array(r.frameToReference)

    def trackFrameBatch(self, references, frames, inits):
        """trackFrame for independent (reference, frame) pairs in the same kernel launches; returns (poses n x 7,
        list of result records)."""
        n = len(frames)
        kfs = (C.c_void_p * n)(*[r.keyframe.h_ for r in references])
        frs = (C.c_void_p * n)(*[f.h_ for f in frames])
        init = np.ascontiguousarray(inits, dtype=np.float64).reshape(n, 7)
        res = (capi.TrackResult * n)()
        check(self.L.lsdhip_tracker_track_batch(self.h_, n, kfs, frs, init.ctypes.data, res))
        for f, r, ref in zip(frames, res, references):
            if not r.diverged:
                f._parent = ref.keyframe
        return np.array([list(r.frameToReference) for r in res]), list(res)

    def evalThroughput(self, references, frames, referenceToFrames, level, repeats=20):
        """measurement hook: the throughput-mode evaluation launch at `level` for n >= 8 jobs; returns (ms per launch, algorithmic
        bytes per launch)"""
        n = len(frames)
        kfs = (C.c_void_p * n)(*[r.keyframe.h_ for r in references])
        frs = (C.c_void_p * n)(*[f.h_ for f in frames])
        T = np.ascontiguousarray(referenceToFrames, dtype=np.float32).reshape(n, 7)
        ms, nb = C.c_double(), C.c_double()
        check(self.L.lsdhip_tracker_eval_throughput(self.h_, n, kfs, frs, T.ctypes.data, int(level), int(repeats), C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def evaluate(self, reference, frame, referenceToFrame, level, a=1.0, b=0.0):
        T = np.ascontiguousarray(referenceToFrame, dtype=np.float32)
        r = capi.ResidualRecord()
        check(self.L.lsdhip_tracker_evaluate(self.h_, reference.keyframe.h_, frame.h_, T.ctypes.data, level, a, b, C.byref(r)), False)
        return r

    def trackFrameOnPermaref(self, permaRef_pos, permaRef_colVar, frame, referenceToFrame):
        pos = np.ascontiguousarray(permaRef_pos, np.float32)
        cv = np.ascontiguousarray(permaRef_colVar, np.float32)
        T = np.ascontiguousarray(referenceToFrame, np.float64)
        r = capi.TrackResult()
        check(self.L.lsdhip_tracker_track_permaref(self.h_, pos.ctypes.data, cv.ctypes.data, len(pos), frame.h_, T.ctypes.data, C.byref(r)))
        self._publish(r)
        return np.array(r.frameToReference)

    def trackFrameOnPermarefBatch(self, clouds, frames, referenceToFrames):
        """clouds: list of (pos n_j x 3, colVar n_j x 2); each tracked against frames[j]; returns (poses, records)"""
        n = len(clouds)
        pos = np.ascontiguousarray(np.concatenate([c[0] for c in clouds]), np.float32)
        cv = np.ascontiguousarray(np.concatenate([c[1] for c in clouds]), np.float32)
        counts = np.array([len(c[0]) for c in clouds], np.int32)
        frs = (C.c_void_p * n)(*[f.h_ for f in frames])
        T = np.ascontiguousarray(referenceToFrames, np.float64).reshape(n, 7)
        res = (capi.TrackResult * n)()
        check(self.L.lsdhip_tracker_track_permaref_batch(self.h_, n, pos.ctypes.data, cv.ctypes.data, counts.ctypes.data, frs, T.ctypes.data, res))
        return np.array([list(r.frameToReference) for r in res]), list(res)

    def checkPermaRefOverlap(self, permaRef_pos, referenceToFrame):
        pos = np.ascontiguousarray(permaRef_pos, np.float32)
        T = np.ascontiguousarray(referenceToFrame, np.float64)
        u = C.c_float()
        check(self.L.lsdhip_tracker_check_overlap(self.h_, pos.ctypes.data, len(pos), T.ctypes.data, C.byref(u)), False)
        self.pointUsage = u.value
        return u.value


class Sim3Tracker:
    """Sim3Tracker (C/Tracking/Sim3Tracker.h:71-187).  Sim3 = (qw,qx,qy,qz,tx,ty,tz,scale)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.L = ctx.L
        h_ = C.c_void_p()
        check(self.L.lsdhip_sim3tracker_create(ctx.h_, C.byref(h_)))
        self.h_ = h_
        self.diverged = False
        self.last = None

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.lsdhip_sim3tracker_destroy(self.h_)
            self.h_ = None

    def setMaxItsPerLvl(self, its):
        a = np.ascontiguousarray(its, np.int32)
        check(self.L.lsdhip_sim3tracker_set_max_its(self.h_, a.ctypes.data))

    def trackFrameSim3(self, keyframe, frame, frameToReference_initialEstimate, startLevel, finalLevel):
        """-> (frameToReference[8], result record); self.diverged mirrors the reference's flag"""
        T = np.ascontiguousarray(frameToReference_initialEstimate, np.float64)
        r = capi.Sim3Result()
        check(self.L.lsdhip_sim3tracker_track(self.h_, keyframe.h_, frame.h_, T.ctypes.data, startLevel, finalLevel, C.byref(r)))
        self.diverged = bool(r.diverged)
        self.last = r
        return np.array(list(r.frameToReference)), r

    def trackFrameSim3Batch(self, keyframes, frames, inits, startLevel, finalLevel):
        """n independent (keyframe, frame, init) jobs in lock step -> (n x 8 poses, list of result records)"""
        n = len(frames)
        kfs = (C.c_void_p * n)(*[k.h_ for k in keyframes])
        frs = (C.c_void_p * n)(*[f.h_ for f in frames])
        T = np.ascontiguousarray(inits, np.float64).reshape(n, 8)
        res = (capi.Sim3Result * n)()
        check(self.L.lsdhip_sim3tracker_track_batch(self.h_, n, kfs, frs, T.ctypes.data, startLevel, finalLevel, res))
        return np.array([list(r.frameToReference) for r in res]), list(res)

    def evaluate(self, keyframe, frame, referenceToFrame, level, a=1.0, b=0.0):
        T = np.ascontiguousarray(referenceToFrame, np.float64)
        r = capi.Sim3EvalRecord()
        check(self.L.lsdhip_sim3tracker_evaluate(self.h_, keyframe.h_, frame.h_, T.ctypes.data, level, a, b, C.byref(r)))
        return r


class DepthMap:
    STAGES = {"observe": 0, "fillholes": 1, "regularize": 2, "regularize_occ": 3, "propagate": 4, "fill_regularize": 5}

    def __init__(self, ctx):
        self.ctx = ctx
        self.L = ctx.L
        h_ = C.c_void_p()
        check(self.L.lsdhip_depth_create(ctx.h_, C.byref(h_)), False)
        self.h_ = h_
        self._keep = []

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "h_", None) and getattr(self.ctx, "h_", None):
            self.L.lsdhip_depth_destroy(self.h_)
        self.h_ = None

    def _arr(self, frames):
        return (C.c_void_p * len(frames))(*[f.h_ for f in frames])

    def isValid(self):
        return bool(self.L.lsdhip_depth_is_valid(self.h_))

    def invalidate(self):
        check(self.L.lsdhip_depth_invalidate(self.h_))

    def reset(self):
        check(self.L.lsdhip_depth_reset(self.h_))

    def initializeFromGTDepth(self, frame):
        self._keep.append(frame)
        check(self.L.lsdhip_depth_init_gt(self.h_, frame.h_), False)

    def initializeRandomly(self, frame):
        self._keep.append(frame)
        check(self.L.lsdhip_depth_init_random(self.h_, frame.h_), False)

    def setFromExistingKF(self, frame):
        self._keep.append(frame)
        check(self.L.lsdhip_depth_set_from_existing(self.h_, frame.h_), False)

    def updateKeyframe(self, referenceFrames):
        frames = list(referenceFrames)
        check(self.L.lsdhip_depth_update(self.h_, self._arr(frames), len(frames)), False)

    @staticmethod
    def updateKeyframeBatch(maps, frames):
        """updateKeyframe of several sequences' maps, one tracked frame each, in shared launches (lsdhip_depth_update_batch)"""
        n = len(maps)
        ma = (C.c_void_p * n)(*[m.h_ for m in maps])
        fa = (C.c_void_p * n)(*[f.h_ for f in frames])
        check(maps[0].L.lsdhip_depth_update_batch(n, ma, fa), False)

    def createKeyFrame(self, new_keyframe):
        self._keep.append(new_keyframe)
        s = C.c_float()
        check(self.L.lsdhip_depth_create_keyframe(self.h_, new_keyframe.h_, C.byref(s)), False)
        return s.value

    def finalizeKeyFrame(self):
        check(self.L.lsdhip_depth_finalize(self.h_), False)

    @staticmethod
    def changeKeyframeBatch(maps, new_keyframes):
        """finalizeKeyFrame() + createKeyFrame(new_keyframes[j]) of several sequences' maps in shared launches
        (lsdhip_depth_change_keyframe_batch); returns the rescale factors"""
        n = len(maps)
        for m, f in zip(maps, new_keyframes):
            m._keep.append(f)
        ma = (C.c_void_p * n)(*[m.h_ for m in maps])
        fa = (C.c_void_p * n)(*[f.h_ for f in new_keyframes])
        s = (C.c_float * n)()
        check(maps[0].L.lsdhip_depth_change_keyframe_batch(n, ma, fa, s), False)
        return [float(v) for v in s]

    def currentDepthMap(self):
        out = np.zeros((self.ctx.h, self.ctx.w), dtype=HYP_DTYPE)
        check(self.L.lsdhip_depth_download(self.h_, out.ctypes.data), False)
        return out

    def setCurrentDepthMap(self, kf, hyp, reactivated=False):
        self._keep.append(kf)
        hyp = np.ascontiguousarray(hyp, dtype=HYP_DTYPE)
        check(self.L.lsdhip_depth_upload(self.h_, kf.h_, hyp.ctypes.data, int(reactivated)), False)

    def stage(self, name, frames=()):
        frames = list(frames)
        self._keep.extend(frames)
        check(self.L.lsdhip_depth_stage(self.h_, self.STAGES[name], self._arr(frames) if frames else None, len(frames)), False)

    def copyRows(self, row0, nrows, packed_dev_ptr, to_map):
        """rows of the 8 hypothesis planes <-> packed device buffer (29 B/px), see lsdhip_depth_copy_rows_dev"""
        check(self.L.lsdhip_depth_copy_rows_dev(self.h_, row0, nrows, C.c_void_p(packed_dev_ptr), int(to_map)), False)

    def copyPlanesToDevice(self, idepth_ptr, var_ptr):
        check(self.L.lsdhip_depth_copy_planes_dev(self.h_, C.c_void_p(idepth_ptr), C.c_void_p(var_ptr)), False)

    def timings(self):
        out = np.zeros(8, np.float32)
        check(self.L.lsdhip_depth_timings(self.h_, out.ctypes.data))
        keys = ["msUpdate", "msCreate", "msFinalize", "msObserve", "msRegularize", "msPropagate", "msFillHoles", "msSetDepth"]
        return dict(zip(keys, out.tolist()))


class SlamLoop:
    """Minimal single-sequence driver: the part of SlamSystem either side of the hot path, with doSlam=false and
    blockUntilMapped=true semantics (C/SlamSystem.cpp:890-1040 trackFrame, :739-828 doMappingIteration, :542-614
    updateKeyframe, :458-490 createNewCurrentKeyframe), and a deterministic keyframe policy (new keyframe every
    `kf_every` frames) instead of the distance/usage score.  Orchestration only — every per-pixel operation is a
    liblsdhip.so call."""

    def __init__(self, ctx, first_image_or_ptr, depth0, kf_every=10, device_frames=False):
        self.ctx = ctx
        self.device_frames = device_frames
        self.kf_every = kf_every
        self.tracker = SE3Tracker(ctx)
        self.tracker.set_maxItsPerLvl([5, 20, 50, 100, 0])  # C/SlamSystem.cpp:80-81
        self.map = DepthMap(ctx)
        self.reference = TrackingReference()
        self.keyframe = self._frame(0, first_image_or_ptr)
        self.keyframe.setDepthFromGroundTruth(depth0)     # SlamSystem::gtDepthInit (SlamSystem.cpp:831-854)
        self.map.initializeFromGTDepth(self.keyframe)
        self.reference.importFrame(self.keyframe)
        self.keyframe.clearDepthHasBeenUpdatedFlag()
        self.last_frameToKF = IDENTITY.copy()
        self.n_since_kf = 0
        self.frame_id = 0
        self.t_track = 0.0
        self.t_map = 0.0
        self.n_track = 0
        self.n_update = 0
        self.evaluations = 0

    def _frame(self, id_, img):
        if self.device_frames:
            return Frame(self.ctx, id_, device_ptr=img)
        return Frame(self.ctx, id_, image=img)

    def step(self, image_or_ptr, clock):
        """track one frame, then one mapping iteration; returns frameToKeyframe"""
        self.frame_id += 1
        t0 = clock()
        frame = self._frame(self.frame_id, image_or_ptr)
        if self.keyframe.stats()["depthHasBeenUpdatedFlag"]:
            self.reference.importFrame(self.keyframe)
            self.keyframe.clearDepthHasBeenUpdatedFlag()
        est = self.tracker.trackFrame(self.reference, frame, self.last_frameToKF)
        t1 = clock()
        self.t_track += t1 - t0
        self.n_track += 1
        self.evaluations += self.tracker.last.numEvaluations
        if self.tracker.diverged:
            raise RuntimeError("tracking diverged at frame %d" % self.frame_id)
        self.n_since_kf += 1
        if self.n_since_kf >= self.kf_every:
            self.map.finalizeKeyFrame()
            self.map.createKeyFrame(frame)
            self.keyframe = frame
            self.reference.importFrame(frame)
            self.keyframe.clearDepthHasBeenUpdatedFlag()
            self.last_frameToKF = IDENTITY.copy()
            self.n_since_kf = 0
            self.new_keyframe = True
        else:
            self.map.updateKeyframe([frame])
            frame.clear_refPixelWasGood()
            self.last_frameToKF = est
            self.n_update += 1
            self.new_keyframe = False
        self.t_map += clock() - t1
        return est
