"""ctypes binding of liblsdhip.so — exactly the entry points include/lsdhip.h declares.

There is no CPU fallback: if the shared library is missing, or a call fails, this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSDHIP_LIB") or os.path.join(_HERE, "liblsdhip.so")   # LSDHIP_LIB: developer builds only
_lib = None

HYP_DTYPE = np.dtype(
    [("isValid", np.uint8), ("_pad", np.uint8, 3), ("blacklisted", np.int32), ("nextStereoFrameMinID", np.float32),
     ("validity_counter", np.int32), ("idepth", np.float32), ("idepth_var", np.float32),
     ("idepth_smoothed", np.float32), ("idepth_var_smoothed", np.float32)])
assert HYP_DTYPE.itemsize == 32


class BuildDefaults(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("ctx_async", "ctx_pipeline", "spec_trials_small", "spec_small_pixels", "spec_trials_mid", "spec_mid_pixels",
                                         "spec_workgroups", "spec_workgroups_above_pixels", "spec_trials_max", "batch_throughput_min_jobs",
                                         "batch_strip_workgroups", "batch_coarse_min_jobs", "batch_coarse_max_pixels",
                                         "batch_coarse_max_points")]


class Params(C.Structure):
    _fields_ = [("minUseGrad", C.c_float), ("cameraPixelNoise2", C.c_float), ("depthSmoothingFactor", C.c_float),
                ("allowNegativeIdepths", C.c_int), ("useSubpixelStereo", C.c_int),
                ("useAffineLightningEstimation", C.c_int)]


class TrackerSettings(C.Structure):
    """lsdhip_tracker_settings = DenseDepthTrackerSettings (C/util/settings.h:355-402)"""
    _fields_ = [("lambdaSuccessFac", C.c_float), ("lambdaFailFac", C.c_float), ("lambdaInitial", C.c_float * 5), ("stepSizeMin", C.c_float * 5),
                ("convergenceEps", C.c_float * 5), ("maxItsPerLvl", C.c_int * 5), ("lambdaInitialTestTrack", C.c_float),
                ("stepSizeMinTestTrack", C.c_float), ("convergenceEpsTestTrack", C.c_float), ("maxItsTestTrack", C.c_float),
                ("huber_d", C.c_float), ("var_weight", C.c_float)]


class TrackResult(C.Structure):
    _fields_ = [("frameToReference", C.c_double * 7), ("pointUsage", C.c_float), ("lastGoodCount", C.c_float),
                ("lastBadCount", C.c_float), ("lastMeanRes", C.c_float), ("lastResidual", C.c_float),
                ("affineEstimation_a", C.c_float), ("affineEstimation_b", C.c_float), ("diverged", C.c_int),
                ("trackingWasGood", C.c_int), ("numEvaluations", C.c_int), ("numWarpUpdates", C.c_int)]


class ResidualRecord(C.Structure):
    _fields_ = [("warped_size", C.c_int), ("goodCount", C.c_float), ("badCount", C.c_float), ("pointUsage", C.c_float),
                ("meanRes", C.c_float), ("retval", C.c_float), ("affine_a_lastIt", C.c_float),
                ("affine_b_lastIt", C.c_float), ("weightedError", C.c_float), ("A", C.c_float * 36), ("b", C.c_float * 6),
                ("lsError", C.c_float), ("num_constraints", C.c_double)]


class Sim3Result(C.Structure):
    _fields_ = [("frameToReference", C.c_double * 8), ("lastResidual", C.c_float), ("lastDepthResidual", C.c_float),
                ("lastPhotometricResidual", C.c_float), ("pointUsage", C.c_float), ("affineEstimation_a", C.c_float),
                ("affineEstimation_b", C.c_float), ("diverged", C.c_int), ("numEvaluations", C.c_int),
                ("lastSim3Hessian", C.c_float * 49)]


class Sim3EvalRecord(C.Structure):
    _fields_ = [("warped_size", C.c_int), ("pointUsage", C.c_float), ("affine_a_lastIt", C.c_float), ("affine_b_lastIt", C.c_float),
                ("sumResD", C.c_float), ("sumResP", C.c_float), ("numTermsD", C.c_int), ("numTermsP", C.c_int),
                ("meanD", C.c_float), ("meanP", C.c_float), ("mean", C.c_float), ("A", C.c_float * 49), ("b", C.c_float * 7),
                ("num_constraints", C.c_double)]


# every symbol declared in include/lsdhip.h: name -> (restype, argtypes)
def _signatures():
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    pvp = C.POINTER(C.c_void_p)
    return {
        "lsdhip_default_params": (None, [C.POINTER(Params)]),
        "lsdhip_build_defaults": (None, [C.POINTER(BuildDefaults)]),
        "lsdhip_ctx_create": (i, [i, i, i, vp, C.POINTER(Params), pvp]),
        "lsdhip_ctx_destroy": (None, [vp]),
        "lsdhip_ctx_stream": (vp, [vp]),
        "lsdhip_ctx_synchronize": (i, [vp]),
        "lsdhip_ctx_set_async": (i, [vp, i]),
        "lsdhip_ctx_lanes_begin": (i, [vp, i]),
        "lsdhip_ctx_lane_select": (i, [vp, i]),
        "lsdhip_ctx_lanes_end": (i, [vp]),
        "lsdhip_ctx_set_pipeline": (i, [vp, i]),
        "lsdhip_ctx_pipeline": (i, [vp]),
        "lsdhip_ctx_map_stream": (vp, [vp]),
        "lsdhip_frame_create_async": (i, [vp, i, vp, pvp]),
        "lsdhip_frame_create_batch": (i, [vp, i, vp, pvp, i, pvp]),
        "lsdhip_ctx_reserve_frames": (i, [vp, i]),
        "lsdhip_frame_relative_pose": (i, [vp, vp, vp]),
        "lsdhip_frame_publish_depth": (i, [vp]),
        "lsdhip_last_error": (C.c_char_p, []),
        "lsdhip_ctx_intrinsics": (i, [vp, i, vp]),
        "lsdhip_frame_create": (i, [vp, i, vp, pvp]),
        "lsdhip_frame_create_from_device": (i, [vp, i, vp, pvp]),
        "lsdhip_frame_destroy": (None, [vp]),
        "lsdhip_frame_id": (i, [vp]),
        "lsdhip_frame_download": (i, [vp, i, i, vp]),
        "lsdhip_frame_set_depth_gt": (i, [vp, vp, f]),
        "lsdhip_frame_set_depth_planes": (i, [vp, vp, vp]),
        "lsdhip_frame_set_maxgrad": (i, [vp, vp]),
        "lsdhip_frame_get_wasgood": (i, [vp, vp]),
        "lsdhip_frame_set_wasgood": (i, [vp, vp]),
        "lsdhip_frame_clear_wasgood": (i, [vp]),
        "lsdhip_frame_set_pose": (i, [vp, vp, vp, f]),
        "lsdhip_frame_get_pose": (i, [vp, vp]),
        "lsdhip_frame_stats": (i, [vp, vp]),
        "lsdhip_frame_set_counters": (i, [vp, i, i, i, i]),
        "lsdhip_frame_depth_updated": (i, [vp]),
        "lsdhip_frame_clear_depth_updated": (i, [vp]),
        "lsdhip_ref_pointcloud": (i, [vp, i, vp, vp, vp, vp]),
        "lsdhip_tracker_create": (i, [vp, pvp]),
        "lsdhip_tracker_destroy": (None, [vp]),
        "lsdhip_tracker_set_max_its": (i, [vp, vp]),
        "lsdhip_tracker_get_settings": (i, [vp, C.POINTER(TrackerSettings)]),
        "lsdhip_tracker_set_settings": (i, [vp, C.POINTER(TrackerSettings)]),
        "lsdhip_depth_copy_rows_batch": (i, [vp, i, vp]),
        "lsdhip_ctx_alloc_dev": (i, [vp, C.c_size_t, vp]),
        "lsdhip_ctx_ipc_export": (i, [vp, vp, vp]),
        "lsdhip_ctx_ipc_open": (i, [vp, vp, vp]),
        "lsdhip_ctx_ipc_close": (i, [vp, vp]),
        "lsdhip_ctx_flag_set": (i, [vp, vp, i]),
        "lsdhip_ctx_flag_wait": (i, [vp, vp, i, vp]),
        "lsdhip_ctx_memset_dev": (i, [vp, vp, i, C.c_size_t]),
        "lsdhip_ctx_read_dev": (i, [vp, vp, vp, C.c_size_t]),
        "lsdhip_ctx_free_dev": (i, [vp, vp]),
        "lsdhip_host_mark": (None, [i]),
        "lsdhip_ctx_aux_begin": (i, [vp]),
        "lsdhip_ctx_aux_end": (i, [vp]),
        "lsdhip_ctx_aux_join": (i, [vp]),
        "lsdhip_ctx_aux_stream": (vp, [vp]),
        "lsdhip_depth_stage_rows": (i, [vp, i, i, i, i]),
        "lsdhip_depth_stage_rows_batch": (i, [vp, i, pvp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "lsdhip_tracker_exec_stats": (i, [vp, vp]),
        "lsdhip_tracker_set_speculation": (i, [vp, i, i]),
        "lsdhip_tracker_set_batch_coarse_min_jobs": (i, [vp, i]),
        "lsdhip_tracker_launch_stats": (i, [vp, vp]),
        "lsdhip_tracker_summary_stats": (i, [vp, vp]),
        "lsdhip_tracker_step_stats": (i, [vp, vp]),
        "lsdhip_tracker_set_enqueue_hook": (i, [vp, vp, vp]),
        "lsdhip_tracker_track": (i, [vp, vp, vp, vp, C.POINTER(TrackResult)]),
        "lsdhip_tracker_track_batch": (i, [vp, i, pvp, pvp, vp, C.POINTER(TrackResult)]),
        "lsdhip_tracker_eval_throughput": (i, [vp, i, pvp, pvp, vp, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "lsdhip_tracker_evaluate": (i, [vp, vp, vp, vp, i, f, f, C.POINTER(ResidualRecord)]),
        "lsdhip_tracker_track_permaref": (i, [vp, vp, vp, i, vp, vp, C.POINTER(TrackResult)]),
        "lsdhip_tracker_track_permaref_batch": (i, [vp, i, vp, vp, vp, pvp, vp, C.POINTER(TrackResult)]),
        "lsdhip_tracker_check_overlap": (i, [vp, vp, i, vp, C.POINTER(C.c_float)]),
        "lsdhip_sim3tracker_create": (i, [vp, pvp]),
        "lsdhip_sim3tracker_destroy": (None, [vp]),
        "lsdhip_sim3tracker_set_max_its": (i, [vp, vp]),
        "lsdhip_sim3tracker_track": (i, [vp, vp, vp, vp, i, i, C.POINTER(Sim3Result)]),
        "lsdhip_sim3tracker_track_batch": (i, [vp, i, pvp, pvp, vp, i, i, C.POINTER(Sim3Result)]),
        "lsdhip_host_se3f_step": (i, [vp, vp, vp]),
        "lsdhip_host_ldlt6": (i, [vp, vp, vp]),
        "lsdhip_host_sim3_step": (i, [vp, vp, vp]),
        "lsdhip_host_ldlt7": (i, [vp, vp, vp]),
        "lsdhip_sim3tracker_evaluate": (i, [vp, vp, vp, vp, i, f, f, C.POINTER(Sim3EvalRecord)]),
        "lsdhip_depth_create": (i, [vp, pvp]),
        "lsdhip_depth_destroy": (None, [vp]),
        "lsdhip_depth_is_valid": (i, [vp]),
        "lsdhip_depth_invalidate": (i, [vp]),
        "lsdhip_depth_reset": (i, [vp]),
        "lsdhip_depth_init_gt": (i, [vp, vp]),
        "lsdhip_depth_init_random": (i, [vp, vp]),
        "lsdhip_depth_set_from_existing": (i, [vp, vp]),
        "lsdhip_depth_update": (i, [vp, pvp, i]),
        "lsdhip_depth_observe_work": (i, [vp, vp]),
        "lsdhip_depth_update_batch": (i, [i, pvp, pvp]),
        "lsdhip_depth_create_keyframe": (i, [vp, vp, C.POINTER(C.c_float)]),
        "lsdhip_depth_change_keyframe_batch": (i, [i, pvp, pvp, C.POINTER(C.c_float)]),
        "lsdhip_depth_finalize": (i, [vp]),
        "lsdhip_depth_download": (i, [vp, vp]),
        "lsdhip_depth_upload": (i, [vp, vp, vp, i]),
        "lsdhip_depth_stage": (i, [vp, i, pvp, i]),
        "lsdhip_depth_copy_planes_dev": (i, [vp, vp, vp]),
        "lsdhip_depth_copy_rows_dev": (i, [vp, i, i, vp, i]),
        "lsdhip_depth_timings": (i, [vp, vp]),
        "lsdhip_ctx_copy_dev": (i, [vp, vp, vp, C.c_size_t]),
        "lsdhip_depth_gpu_times": (i, [vp, vp, vp]),
        "lsdhip_depth_observe_time": (i, [vp, vp, vp]),
        "lsdhip_prof_enable": (i, [vp, i]),
        "lsdhip_prof_read": (i, [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
        "lsdhip_ctx_batch_prof_read": (i, [vp, vp, vp, vp, vp]),
        "lsdhip_prof_reset": (i, [vp]),
    }


EXPORTED_SYMBOLS = sorted(_signatures().keys())


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("liblsdhip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` — "
                           "there is no CPU fallback for the hot path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in _signatures().items():
        fn = getattr(L, name)  # AttributeError if the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class LsdHipError(RuntimeError):
    pass


def check(rc, allow_positive=True):
    if rc < 0 or (rc > 0 and not allow_positive):
        msg = lib().lsdhip_last_error()
        raise LsdHipError("liblsdhip call failed (%d): %s" % (rc, msg.decode() if msg else ""))
    return rc
