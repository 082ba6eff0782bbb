"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liblsd_oracle*.so and of oracle/_ref (the reference's
own sources behind the same entry points; see oracle/lsd_oracle.hpp for what is pinned and what is not).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

HYP_DTYPE = np.dtype(
    [("isValid", np.uint8), ("_pad", np.uint8, 3), ("blacklisted", np.int32), ("nextStereoFrameMinID", np.float32),
     ("validity_counter", np.int32), ("idepth", np.float32), ("idepth_var", np.float32),
     ("idepth_smoothed", np.float32), ("idepth_var_smoothed", np.float32)])
assert HYP_DTYPE.itemsize == 32


class Params(C.Structure):
    _fields_ = [("minUseGrad", C.c_float), ("cameraPixelNoise2", C.c_float), ("depthSmoothingFactor", C.c_float),
                ("allowNegativeIdepths", C.c_int), ("useSubpixelStereo", C.c_int), ("multiThreading", C.c_int),
                ("useAffineLightningEstimation", C.c_int), ("KFDistWeight", C.c_float), ("KFUsageWeight", C.c_float)]


class TrackResult(C.Structure):
    _fields_ = [("frameToRef", C.c_double * 7), ("pointUsage", C.c_float), ("lastGoodCount", C.c_float),
                ("lastBadCount", C.c_float), ("lastMeanRes", C.c_float), ("lastResidual", C.c_float),
                ("affine_a", C.c_float), ("affine_b", C.c_float), ("diverged", C.c_int), ("trackingWasGood", C.c_int),
                ("numEvaluations", C.c_int), ("numWarpUpdates", C.c_int)]


class ResidualRecord(C.Structure):
    _fields_ = [("warped_size", C.c_int), ("goodCount", C.c_float), ("badCount", C.c_float), ("pointUsage", C.c_float),
                ("meanRes", C.c_float), ("retval", C.c_float), ("affine_a_lastIt", C.c_float),
                ("affine_b_lastIt", C.c_float), ("weightedError", C.c_float), ("A", C.c_float * 36), ("b", C.c_float * 6),
                ("lsError", C.c_float), ("num_constraints", C.c_double)]


class Sim3Result(C.Structure):
    _fields_ = [("frameToRef", C.c_double * 8), ("lastResidual", C.c_float), ("lastDepthResidual", C.c_float),
                ("lastPhotometricResidual", C.c_float), ("pointUsage", C.c_float), ("affine_a", C.c_float), ("affine_b", C.c_float),
                ("diverged", C.c_int), ("numEvaluations", C.c_int), ("hessian", C.c_float * 49)]


class Sim3EvalRecord(C.Structure):
    _fields_ = [("warped_size", C.c_int), ("pointUsage", C.c_float), ("affine_a_lastIt", C.c_float), ("affine_b_lastIt", C.c_float),
                ("sumResD", C.c_float), ("sumResP", C.c_float), ("numTermsD", C.c_int), ("numTermsP", C.c_int),
                ("meanD", C.c_float), ("meanP", C.c_float), ("mean", C.c_float), ("A", C.c_float * 49), ("b", C.c_float * 7),
                ("num_constraints", C.c_double)]


def build(force=False):
    """Compile both oracle builds (parity + timing) with the committed Makefile."""
    need = force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("liblsd_oracle.so", "liblsd_oracle_fast.so"))
    if need or True:
        subprocess.check_call(["make", "-C", _HERE, "-s"])


REF_DIR = os.path.join(_HERE, "_ref")


def have_ref():
    """oracle/_ref/liblsd_ref_{sse,scalar}.so: the reference's own hot-path sources compiled against the stand-in
    dependency headers (oracle/ref/).  Built by `make -C oracle ref` where /root/reference exists; git-ignored, travels
    with the gpurun snapshot."""
    return all(os.path.exists(os.path.join(REF_DIR, "liblsd_ref_%s.so" % k)) for k in ("sse", "scalar"))


def build_ref():
    """Compile oracle/_ref from the reference sources where they lie (only possible where /root/reference exists)."""
    if os.path.isdir("/root/reference/lsd_slam_core/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return have_ref()


def ref_timing_variant():
    """Which timing build of the reference (oracle/_ref/liblsd_ref_sse_o3v{3,4}.so: -O3 -DENABLE_SSE -DNDEBUG, contraction at the
    compiler default — the way lsd_slam_core/CMakeLists.txt builds it, with a portable -march level instead of -march=native) this
    host can run: the highest level its cpuid supports.  Returns (lib key, flags string) or (None, None)."""
    flags = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = line
                break
    except Exception:
        pass
    have = set(flags.split())
    v3 = {"avx2", "fma", "bmi2", "movbe", "f16c"} <= have
    v4 = v3 and {"avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl"} <= have
    for key, ok, march in (("sse_o3v4", v4, "x86-64-v4"), ("sse_o3v3", v3, "x86-64-v3")):
        if ok and os.path.exists(os.path.join(REF_DIR, "liblsd_ref_%s.so" % key)):
            return key, "-O3 -march=%s -DENABLE_SSE -DNDEBUG (floating-point contraction at the compiler default)" % march
    return None, None


def build_native():
    """-O3 -march=native timing build for the host this runs on (falls back to the portable -march=x86-64-v3 build)"""
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
        return os.path.exists(os.path.join(_HERE, "liblsd_oracle_native.so"))
    except Exception:
        return False


def lib(fast=False, ref=None, native=False):
    """fast: the -O3 timing build of the oracle.  ref = "sse" | "scalar": the REFERENCE itself (oracle/_ref) behind the
    same entry points — every class below works on either library."""
    if ref is not None:
        name = os.path.join("_ref", "liblsd_ref_%s.so" % ref)
    elif native:
        name = "liblsd_oracle_native.so"
    else:
        name = "liblsd_oracle_fast.so" if fast else "liblsd_oracle.so"
    if name in _LIBS:
        return _LIBS[name]
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        if ref is not None:
            raise FileNotFoundError(path + " (make -C oracle ref, needs /root/reference)")
        build()
    L = C.CDLL(path)
    vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    sig = {
        "orc_default_params": (None, [C.POINTER(Params)]),
        "orc_frame_create": (vp, [i, i, i, fp, u8p]),
        "orc_frame_destroy": (None, [vp]),
        "orc_frame_get": (i, [vp, i, i, fp]),
        "orc_frame_intrinsics": (None, [vp, i, fp]),
        "orc_frame_set_sse_pyramid": (None, [vp, i]),
        "orc_frame_set_depth_gt": (None, [vp, fp, f, f]),
        "orc_frame_set_depth_planes": (None, [vp, fp, fp]),
        "orc_frame_get_wasgood": (i, [vp, u8p]),
        "orc_frame_set_wasgood": (None, [vp, u8p]),
        "orc_frame_clear_wasgood": (None, [vp]),
        "orc_frame_set_maxgrad": (None, [vp, fp]),
        "orc_frame_set_pose": (None, [vp, dp, vp, f]),
        "orc_frame_get_pose": (None, [vp, dp]),
        "orc_frame_stats": (None, [vp, fp]),
        "orc_frame_set_counters": (None, [vp, i, i, i, i]),
        "orc_frame_stereo_precomp": (None, [vp, fp]),
        "orc_ref_create": (vp, []),
        "orc_ref_destroy": (None, [vp]),
        "orc_ref_import": (None, [vp, vp]),
        "orc_ref_pointcloud": (i, [vp, i, vp, vp, vp, vp]),
        "orc_tracker_create": (vp, [i, i, fp, C.POINTER(Params)]),
        "orc_tracker_destroy": (None, [vp]),
        "orc_tracker_set_mode": (None, [vp, i]),
        "orc_tracker_set_max_its": (None, [vp, ip]),
        "orc_tracker_track": (None, [vp, vp, vp, dp, C.POINTER(TrackResult)]),
        "orc_tracker_evaluate": (None, [vp, vp, vp, fp, i, f, f, C.POINTER(ResidualRecord)]),
        "orc_tracker_buffer": (i, [vp, i, vp]),
        "orc_tracker_track_permaref": (None, [vp, fp, fp, i, vp, dp, C.POINTER(TrackResult)]),
        "orc_tracker_check_overlap": (f, [vp, fp, i, vp, dp]),
        "orc_se3_exp_f": (None, [fp, fp]),
        "orc_se3_exp_d": (None, [dp, dp]),
        "orc_se3_log_d": (None, [dp, dp]),
        "orc_se3_mul_d": (None, [dp, dp, dp]),
        "orc_se3_inv_d": (None, [dp, dp]),
        "orc_ldlt6_solve": (None, [fp, fp, fp]),
        "orc_depth_create": (vp, [i, i, fp, C.POINTER(Params)]),
        "orc_depth_destroy": (None, [vp]),
        "orc_depth_set_threads": (None, [vp, i]),
        "orc_depth_init_gt": (None, [vp, vp]),
        "orc_depth_init_random": (None, [vp, vp]),
        "orc_depth_get": (None, [vp, vp]),
        "orc_depth_set": (None, [vp, vp, vp, i]),
        "orc_depth_update": (None, [vp, C.POINTER(vp), i]),
        "orc_depth_create_keyframe": (None, [vp, vp]),
        "orc_depth_last_rescale": (f, [vp]),
        "orc_depth_finalize": (None, [vp]),
        "orc_depth_set_from_existing": (None, [vp, vp]),
        "orc_frame_take_reactivation": (None, [vp, vp]),
        "orc_frame_set_depth_from_map": (None, [vp, vp]),
        "orc_depth_stage": (None, [vp, i, C.POINTER(vp), i]),
        "orc_depth_line_stereo": (None, [vp, vp, i, i, f, f, f, fp]),
        "orc_sim3tracker_create": (vp, [i, i, fp, C.POINTER(Params)]),
        "orc_sim3tracker_destroy": (None, [vp]),
        "orc_sim3tracker_set_mode": (None, [vp, i]),
        "orc_sim3tracker_set_max_its": (None, [vp, ip]),
        "orc_sim3tracker_track": (None, [vp, vp, vp, dp, i, i, C.POINTER(Sim3Result)]),
        "orc_sim3tracker_evaluate": (None, [vp, vp, vp, dp, i, f, f, C.POINTER(Sim3EvalRecord)]),
        "orc_sim3_exp": (None, [dp, dp]),
        "orc_now_seconds": (d, []),
    }
    for name_, (res, args) in sig.items():
        if ref is not None and not hasattr(L, name_):
            continue          # the reference library exports the Frame / TrackingReference / SE3Tracker / DepthMap entry points only
        fn = getattr(L, name_)
        fn.restype = res
        fn.argtypes = args
    _LIBS[name] = L
    return L


def default_params(L=None):
    L = L or lib()
    p = Params()
    L.orc_default_params(C.byref(p))
    return p


SCALAR, SSE, SSE_EXACT_RCP = 0, 1, 2


class Frame:
    def __init__(self, id_, img, K, L=None):
        self.L = L or lib()
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self.h, self.w = img.shape
        self.id = id_
        self.K = np.ascontiguousarray(K, dtype=np.float32)
        self.h_ = self.L.orc_frame_create(id_, self.w, self.h, self.K, img)

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_frame_destroy(self.h_)
            self.h_ = None

    def dims(self, level):
        return self.w >> level, self.h >> level

    def plane(self, what, level=0):
        wl, hl = self.dims(level)
        names = {"image": 0, "gradients": 1, "maxGradients": 2, "idepth": 3, "idepthVar": 4}
        k = names[what]
        out = np.zeros((hl, wl, 4) if k == 1 else (hl, wl), dtype=np.float32)
        rc = self.L.orc_frame_get(self.h_, k, level, out)
        if rc != 0:
            raise RuntimeError("plane %s not available" % what)
        return out

    def intrinsics(self, level):
        out = np.zeros(8, dtype=np.float32)
        self.L.orc_frame_intrinsics(self.h_, level, out)
        return out

    def set_depth_gt(self, depth, cov_scale=1.0, minUseGrad=5.0):
        self.L.orc_frame_set_depth_gt(self.h_, np.ascontiguousarray(depth, dtype=np.float32), cov_scale, minUseGrad)

    def set_depth_planes(self, idepth, var):
        self.L.orc_frame_set_depth_planes(self.h_, np.ascontiguousarray(idepth, dtype=np.float32),
                                          np.ascontiguousarray(var, dtype=np.float32))

    def wasgood(self):
        wl, hl = self.dims(1)
        out = np.zeros((hl, wl), dtype=np.uint8)
        if not self.L.orc_frame_get_wasgood(self.h_, out):
            return None
        return out

    def set_wasgood(self, m):
        self.L.orc_frame_set_wasgood(self.h_, np.ascontiguousarray(m, dtype=np.uint8))

    def clear_wasgood(self):
        self.L.orc_frame_clear_wasgood(self.h_)

    def set_maxgrad(self, plane):
        self.L.orc_frame_set_maxgrad(self.h_, np.ascontiguousarray(plane, dtype=np.float32))

    def set_pose(self, sim3, parent, initialTrackedResidual=0.0):
        self.L.orc_frame_set_pose(self.h_, np.ascontiguousarray(sim3, dtype=np.float64), parent.h_ if parent else None,
                                  initialTrackedResidual)

    def pose(self):
        out = np.zeros(8, dtype=np.float64)
        self.L.orc_frame_get_pose(self.h_, out)
        return out

    def stats(self):
        out = np.zeros(8, dtype=np.float32)
        self.L.orc_frame_stats(self.h_, out)
        keys = ["initialTrackedResidual", "meanIdepth", "numPoints", "numFramesTrackedOnThis", "numMappedOnThis",
                "numMappedOnThisTotal", "depthHasBeenUpdatedFlag", "numMappablePixels"]
        return dict(zip(keys, out.tolist()))

    def set_counters(self, tracked, mapped, mappedTotal, depthUpdated):
        self.L.orc_frame_set_counters(self.h_, tracked, mapped, mappedTotal, int(depthUpdated))

    def stereo_precomp(self):
        out = np.zeros(27, dtype=np.float32)
        self.L.orc_frame_stereo_precomp(self.h_, out)
        return out


class TrackingReference:
    def __init__(self, L=None):
        self.L = L or lib()
        self.h_ = self.L.orc_ref_create()
        self.frame = None

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_ref_destroy(self.h_)
            self.h_ = None

    def import_frame(self, frame):
        self.frame = frame
        self.L.orc_ref_import(self.h_, frame.h_)

    def pointcloud(self, level):
        wl, hl = self.frame.dims(level)
        n_max = wl * hl
        pos = np.zeros((n_max, 3), np.float32)
        cv = np.zeros((n_max, 2), np.float32)
        gr = np.zeros((n_max, 2), np.float32)
        idx = np.zeros(n_max, np.int32)
        n = self.L.orc_ref_pointcloud(self.h_, level, pos.ctypes.data, cv.ctypes.data, gr.ctypes.data, idx.ctypes.data)
        return pos[:n].copy(), cv[:n].copy(), gr[:n].copy(), idx[:n].copy()


class SE3Tracker:
    def __init__(self, w, h, K, params=None, mode=SSE, L=None):
        self.L = L or lib()
        self.params = params or default_params(self.L)
        self.h_ = self.L.orc_tracker_create(w, h, np.ascontiguousarray(K, dtype=np.float32), C.byref(self.params))
        self.L.orc_tracker_set_mode(self.h_, mode)

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_tracker_destroy(self.h_)
            self.h_ = None

    def set_max_its(self, its):
        self.L.orc_tracker_set_max_its(self.h_, np.ascontiguousarray(its, dtype=np.int32))

    def track(self, ref, frame, init_frameToRef):
        r = TrackResult()
        self.L.orc_tracker_track(self.h_, ref.h_, frame.h_, np.ascontiguousarray(init_frameToRef, dtype=np.float64), C.byref(r))
        return r

    def evaluate(self, ref, frame, refToFrame7, level, a=1.0, b=0.0):
        r = ResidualRecord()
        self.L.orc_tracker_evaluate(self.h_, ref.h_, frame.h_, np.ascontiguousarray(refToFrame7, dtype=np.float32), level,
                                    a, b, C.byref(r))
        return r

    def buffer(self, which):
        names = ["x", "y", "z", "dx", "dy", "residual", "d", "idepthVar", "weight_p"]
        k = names.index(which)
        n = self.L.orc_tracker_buffer(self.h_, k, None)
        out = np.zeros(n, np.float32)
        self.L.orc_tracker_buffer(self.h_, k, out.ctypes.data)
        return out

    def track_permaref(self, pos, colvar, frame, refToFrame):
        r = TrackResult()
        pos = np.ascontiguousarray(pos, np.float32)
        colvar = np.ascontiguousarray(colvar, np.float32)
        self.L.orc_tracker_track_permaref(self.h_, pos, colvar, len(pos), frame.h_,
                                          np.ascontiguousarray(refToFrame, np.float64), C.byref(r))
        return r

    def check_overlap(self, pos, ref_frame, refToFrame):
        pos = np.ascontiguousarray(pos, np.float32)
        return self.L.orc_tracker_check_overlap(self.h_, pos, len(pos), ref_frame.h_,
                                                np.ascontiguousarray(refToFrame, np.float64))


class Sim3Tracker:
    """C/Tracking/Sim3Tracker.{h,cpp} restated (oracle/orc_sim3.cpp).  Sim3 = double[8] (qw,qx,qy,qz,tx,ty,tz,scale)."""

    def __init__(self, w, h, K, params=None, mode=SSE, L=None):
        self.L = L or lib()
        self.params = params or default_params(self.L)
        self.h_ = self.L.orc_sim3tracker_create(w, h, np.ascontiguousarray(K, dtype=np.float32), C.byref(self.params))
        self.L.orc_sim3tracker_set_mode(self.h_, mode)

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_sim3tracker_destroy(self.h_)
            self.h_ = None

    def set_max_its(self, its):
        self.L.orc_sim3tracker_set_max_its(self.h_, np.ascontiguousarray(its, dtype=np.int32))

    def track(self, ref, frame, init_frameToRef8, startLevel, finalLevel):
        r = Sim3Result()
        self.L.orc_sim3tracker_track(self.h_, ref.h_, frame.h_, np.ascontiguousarray(init_frameToRef8, dtype=np.float64), startLevel,
                                     finalLevel, C.byref(r))
        return r

    def evaluate(self, ref, frame, refToFrame8, level, a=1.0, b=0.0):
        r = Sim3EvalRecord()
        self.L.orc_sim3tracker_evaluate(self.h_, ref.h_, frame.h_, np.ascontiguousarray(refToFrame8, dtype=np.float64), level, a, b,
                                        C.byref(r))
        return r


def sim3_exp(a7):
    out = np.zeros(8)
    lib().orc_sim3_exp(np.ascontiguousarray(a7, dtype=np.float64), out)
    return out


def sim3_inv(T):
    """inverse of (qw,qx,qy,qz,tx,ty,tz,s): p' = s R p + t"""
    q = np.array([T[0], -T[1], -T[2], -T[3]])
    s = 1.0 / T[7]
    R = quat_to_rot(q)
    return np.concatenate([q, -(R @ np.asarray(T[4:7])) * s, [s]])


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class DepthMap:
    def __init__(self, w, h, K, params=None, L=None, threads=4):
        self.L = L or lib()
        self.w, self.h = w, h
        self.params = params or default_params(self.L)
        self.h_ = self.L.orc_depth_create(w, h, np.ascontiguousarray(K, dtype=np.float32), C.byref(self.params))
        self.L.orc_depth_set_threads(self.h_, threads)
        self._keep = []

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_depth_destroy(self.h_)
            self.h_ = None

    def init_gt(self, frame):
        self._keep.append(frame)
        self.L.orc_depth_init_gt(self.h_, frame.h_)

    def init_random(self, frame):
        self._keep.append(frame)
        self.L.orc_depth_init_random(self.h_, frame.h_)

    def get(self):
        out = np.zeros((self.h, self.w), dtype=HYP_DTYPE)
        self.L.orc_depth_get(self.h_, out.ctypes.data)
        return out

    def set(self, kf, hyp, reactivated=False):
        self._keep.append(kf)
        hyp = np.ascontiguousarray(hyp, dtype=HYP_DTYPE)
        self.L.orc_depth_set(self.h_, kf.h_, hyp.ctypes.data, int(reactivated))

    def _frames(self, frames):
        arr = (C.c_void_p * len(frames))(*[f.h_ for f in frames])
        return arr

    def update(self, frames):
        self.L.orc_depth_update(self.h_, self._frames(frames), len(frames))

    def create_keyframe(self, frame):
        self._keep.append(frame)
        self.L.orc_depth_create_keyframe(self.h_, frame.h_)
        return self.L.orc_depth_last_rescale(self.h_)

    def finalize(self):
        self.L.orc_depth_finalize(self.h_)

    def set_from_existing(self, frame):
        self._keep.append(frame)
        self.L.orc_depth_set_from_existing(self.h_, frame.h_)

    def line_stereo(self, ref_frame, x, y, min_idepth, prior_idepth, max_idepth):
        """makeAndCheckEPL + doLineStereo for one pixel (no map update): (isGood, epx, epy, error, idepth, var, eplLength)"""
        out = np.zeros(7, np.float32)
        self.L.orc_depth_line_stereo(self.h_, ref_frame.h_, int(x), int(y), min_idepth, prior_idepth, max_idepth, out)
        return out

    def stage(self, name, frames=()):
        k = {"observe": 0, "fillholes": 1, "regularize": 2, "regularize_occ": 3, "propagate": 4}[name]
        frames = list(frames)
        self._keep.extend(frames)
        self.L.orc_depth_stage(self.h_, k, self._frames(frames) if frames else None, len(frames))


def se3_exp(a):
    out = np.zeros(7)
    lib().orc_se3_exp_d(np.ascontiguousarray(a, np.float64), out)
    return out


def se3_log(p):
    out = np.zeros(6)
    lib().orc_se3_log_d(np.ascontiguousarray(p, np.float64), out)
    return out


def se3_mul(a, b):
    out = np.zeros(7)
    lib().orc_se3_mul_d(np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64), out)
    return out


def se3_inv(a):
    out = np.zeros(7)
    lib().orc_se3_inv_d(np.ascontiguousarray(a, np.float64), out)
    return out
